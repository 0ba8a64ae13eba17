#!/usr/bin/env python3
"""Host-side cost of one device-pointer call, and what HIP graph capture of a run of small calls buys.
Small batches (a wavefront's late bounces) are launch-bound: DESIGN.md §6 'throughput vs batch size'."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from mitsuba_customization_amd import host, synth
    with host.MerlHip(0) as g:
        t = g.upload_merl(synth.make_table("ggx_tab", seed=1))
        for n in (1 << 10, 1 << 14, 1 << 17):
            wi, wo, u = g.generate_pairs(1, 0, n)
            out = g.eval_sample(wi, wo, u, material=t)
            g.synchronize()
            calls = 2000
            t0 = time.perf_counter()
            for _ in range(calls):
                g.eval_sample(wi, wo, u, material=t, out=out)
            t_issue = time.perf_counter() - t0
            g.synchronize()
            t_done = time.perf_counter() - t0
            ref = [o.clone() for o in out]

            # the same run of calls captured once into a HIP graph and replayed
            per_graph = 50
            side = torch.cuda.Stream()
            graph = torch.cuda.CUDAGraph()
            for o in out:
                o.zero_()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(per_graph):
                    g.eval_sample(wi, wo, u, material=t, out=out)
            graph.replay(); torch.cuda.synchronize()
            same = all(torch.equal(a, b) for a, b in zip(out, ref))
            t0 = time.perf_counter()
            for _ in range(calls // per_graph):
                graph.replay()
            torch.cuda.synchronize()
            t_graph = time.perf_counter() - t0
            print(json.dumps({"units_per_call": n, "calls": calls,
                              "us_per_call_issue": round(t_issue / calls * 1e6, 2),
                              "us_per_call_complete": round(t_done / calls * 1e6, 2),
                              "us_per_call_graph_replay": round(t_graph / calls * 1e6, 2),
                              "graph_output_identical": same}), flush=True)


if __name__ == "__main__":
    main()
