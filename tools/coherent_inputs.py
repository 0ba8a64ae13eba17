#!/usr/bin/env python3
"""What the fused kernel does when the table traffic is cache-served: the same 64M-unit launch on inputs of
decreasing incoherence (all units identical -> 4K distinct pairs repeated -> fully random).  python tools/coherent_inputs.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mitsuba_customization_amd import host, synth

gpu = host.MerlHip(0)
mid = gpu.upload_merl(synth.make_table("ggx_tab", 0))
n = 64 << 20
wi, wo, u = gpu.generate_pairs(0x5EED, 0, n)
out = (torch.empty((n, 3), dtype=torch.float32, device="cuda"), torch.empty((n,), dtype=torch.float32, device="cuda"),
       torch.empty((n, 3), dtype=torch.float32, device="cuda"), torch.empty((n,), dtype=torch.float32, device="cuda"),
       torch.empty((n, 3), dtype=torch.float32, device="cuda"))
res = {}
for name, period in (("random (bench)", n), ("period 2^20", 1 << 20), ("period 2^16", 1 << 16), ("period 4096", 4096), ("all identical", 1)):
    if period < n:
        a, b, c = wi[:period].repeat(n // period, 1), wo[:period].repeat(n // period, 1), u[:period].repeat(n // period, 1)
    else:
        a, b, c = wi, wo, u
    for _ in range(2):
        gpu.eval_sample(a, b, c, material=mid, out=out)
    torch.cuda.synchronize()
    gpu.timer_start()
    for _ in range(5):
        gpu.eval_sample(a, b, c, material=mid, out=out)
    ms = gpu.timer_stop() / 5
    res[name] = {"ms": round(ms, 3), "G_units_per_s": round(n / ms / 1e6, 2), "stream_GBps": round(76 * n / ms / 1e6, 1)}
    del a, b, c
print(json.dumps(res, indent=1))
