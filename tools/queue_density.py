#!/usr/bin/env python3
"""Throughput of mrl_eval_sample_queue against the share of path slots that are queued and the queue's order.

A wavefront integrator can either keep its path state dense (compact the arrays every bounce, then call the
whole-array entry point) or leave the state in place and pass a queue of live slots.  The queue costs nothing to
build but reads whole 128-B lines of the slot arrays for 12-B payloads once it gets sparse; this measures where
the break-even is.  One JSON line per (density, order)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from mitsuba_customization_amd import host, synth
    n = 64 << 20
    with host.MerlHip(0) as g:
        t = g.upload_merl(synth.make_table("ggx_tab", seed=1))
        wi, wo, u = g.generate_pairs(0x5EED, 0, n)
        out = g.eval_sample(wi, wo, u, material=t)
        g.synchronize()

        def timed(fn, reps=10):
            fn(); g.synchronize()
            g.timer_start()
            for _ in range(reps):
                fn()
            return g.timer_stop() / reps

        ms = timed(lambda: g.eval_sample(wi, wo, u, material=t, out=out))
        print(json.dumps({"call": "eval_sample_batch", "units": n, "ms": round(ms, 4), "Gunits_per_s": round(n / ms / 1e6, 2)}), flush=True)
        gen = torch.Generator(device="cuda").manual_seed(1)
        for density in (1.0, 0.5, 0.25, 0.125, 0.03125):
            k = int(n * density)
            keep = torch.rand(n, device="cuda", generator=gen) < density if density < 1.0 else torch.ones(n, dtype=torch.bool, device="cuda")
            ascending = keep.nonzero().flatten().to(torch.int32)
            k = int(ascending.numel())
            count = torch.tensor([k], dtype=torch.int32, device="cuda")
            shuffled = ascending[torch.randperm(k, device="cuda", generator=gen)].contiguous()
            for order, q in (("ascending", ascending), ("shuffled", shuffled)):
                ms = timed(lambda: g.eval_sample_queue(wi, wo, u, q, count, material=t, out=out), reps=5)
                print(json.dumps({"call": "eval_sample_queue", "density": density, "order": order, "units": k, "ms": round(ms, 4),
                                  "Gunits_per_s": round(k / ms / 1e6, 2)}), flush=True)
            del keep, ascending, shuffled


if __name__ == "__main__":
    main()
