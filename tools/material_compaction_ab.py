#!/usr/bin/env python3
"""BASELINE configs[3] (16 MERL materials mixed in one batch): the fused mixed-material launch against per-material
compaction (mrl_partition_by_material, then one single-material queue launch per material).  One JSON object."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from mitsuba_customization_amd import host, synth
    n = 64 << 20
    with host.MerlHip(0) as g:
        distinct = [synth.make_table("ggx_tab", s) for s in range(16)]
        ids = [g.upload_merl(t) for t in distinct]
        wi, wo, u = g.generate_pairs(0x5EED, 0, n)
        mat = g.generate_materials(0x5EED, 0, n, len(ids))
        out = g.eval_sample(wi, wo, u, mat=mat)
        g.synchronize()

        def timed(fn, reps=5):
            fn(); g.synchronize()
            g.timer_start()
            for _ in range(reps):
                fn()
            return g.timer_stop() / reps

        res = {"units": n, "materials": len(ids)}
        res["fused_ms"] = timed(lambda: g.eval_sample(wi, wo, u, mat=mat, out=out))
        res["partition_ms"] = timed(lambda: g.partition_by_material(mat))
        queue, offsets, counts = g.partition_by_material(mat)
        off = offsets.cpu().tolist()

        def per_material():
            for m in ids:
                g.eval_sample_queue(wi, wo, u, queue[off[m]:off[m + 1]], counts[m:m + 1], material=m, out=out)

        res["per_material_queues_ms"] = timed(per_material)
        res["compacted_total_ms"] = res["partition_ms"] + res["per_material_queues_ms"]
        res["fused_Gunits_per_s"] = n / res["fused_ms"] / 1e6
        res["compacted_Gunits_per_s"] = n / res["compacted_total_ms"] / 1e6
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}))


if __name__ == "__main__":
    main()
