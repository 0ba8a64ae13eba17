#!/usr/bin/env python3
"""Throughput of the fused eval+sample launch versus batch size (device-resident inputs).  python tools/size_sweep.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mitsuba_customization_amd import host, synth

gpu = host.MerlHip(0)
mid = gpu.upload_merl(synth.make_table("ggx_tab", 0))
nmax = 64 << 20
wi, wo, u = gpu.generate_pairs(0x5EED, 0, nmax)
out = (torch.empty((nmax, 3), dtype=torch.float32, device="cuda"), torch.empty((nmax,), dtype=torch.float32, device="cuda"),
       torch.empty((nmax, 3), dtype=torch.float32, device="cuda"), torch.empty((nmax,), dtype=torch.float32, device="cuda"),
       torch.empty((nmax, 3), dtype=torch.float32, device="cuda"))
res = {}
for lg in (10, 12, 14, 16, 18, 20, 22, 24, 26):
    n = 1 << lg
    o = tuple(t[:n] for t in out)
    call = lambda: gpu.eval_sample(wi[:n], wo[:n], u[:n], material=mid, out=o)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    reps = max(3, min(200, (1 << 27) // n))
    gpu.timer_start()
    for _ in range(reps):
        call()
    ms = gpu.timer_stop() / reps
    res[f"2^{lg}"] = {"us_per_launch": round(ms * 1e3, 2), "G_units_per_s": round(n / ms / 1e6, 3)}
print(json.dumps(res, indent=1))
