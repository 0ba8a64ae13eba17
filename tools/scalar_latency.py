#!/usr/bin/env python3
"""Latency of small batches through the C ABI (what a scalar plugin call costs).  python tools/scalar_latency.py"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mitsuba_customization_amd import host, synth

gpu = host.MerlHip(0)
gpu.use_own_stream()
mid = gpu.upload_merl(synth.make_table("ggx_tab", 0))
L, ctx = gpu._lib, gpu._ctx
p = C.c_void_p()
assert L.mrl_host_alloc(ctx, 4096, C.byref(p)) == 0
buf = (C.c_float * 1024).from_address(p.value)
buf[0:8] = [0.3, 0.1, 0.9, -0.2, 0.4, 0.8, 0.3, 0.7]
base = p.value
res = {}
for n in (1, 64, 4096):
    for name, call in (("eval", lambda: L.mrl_eval_batch(ctx, base, base + 12, None, mid, 1, base + 64)),
                       ("eval_sample", lambda: L.mrl_eval_sample_batch(ctx, base, base + 12, base + 24, None, mid, 1, base + 64, base + 80, base + 96, base + 112, base + 128))):
        if n > 1 and name == "eval_sample":
            continue
        for _ in range(50):
            call(); L.mrl_synchronize(ctx)
        t0 = time.perf_counter()
        reps = 2000
        for _ in range(reps):
            call(); L.mrl_synchronize(ctx)
        res[f"{name}_n1_pinned_us"] = (time.perf_counter() - t0) / reps * 1e6
    break
# pageable host arrays (staged): n = 1 and n = 4096
for n in (1, 4096):
    wi = np.tile(np.array([[0.3, 0.1, 0.9]], np.float32), (n, 1)); wo = np.tile(np.array([[-0.2, 0.4, 0.8]], np.float32), (n, 1))
    out = np.empty((n, 3), np.float32)
    for _ in range(20):
        gpu.eval(wi, wo, material=mid, out=out)
    t0 = time.perf_counter()
    for _ in range(500):
        gpu.eval(wi, wo, material=mid, out=out)
    res[f"eval_n{n}_pageable_us"] = (time.perf_counter() - t0) / 500 * 1e6
print(json.dumps(res, indent=1))
