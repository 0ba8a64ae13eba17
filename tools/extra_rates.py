#!/usr/bin/env python3
"""Side measurements quoted in DESIGN.md (not the bench line): eval-only / sample-only rates, the
PCIe-inclusive rate of the host-pointer path, scalar plugin-call latency.   python tools/extra_rates.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mitsuba_customization_amd import host, synth

gpu = host.MerlHip(0)
mid = gpu.upload_merl(synth.make_table("ggx_tab", 0))
n = 64 << 20
wi, wo, u = gpu.generate_pairs(0x5EED, 0, n)
res = {}


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    gpu.timer_start()
    for _ in range(reps):
        fn()
    return gpu.timer_stop() / reps


o_rgb = torch.empty((n, 3), dtype=torch.float32, device="cuda")
ms = timed(lambda: gpu.eval(wi, wo, material=mid, out=o_rgb))
res["eval_only"] = {"ms": ms, "Meval_per_s": n / ms / 1e3, "stream_GBps": 36 * n / ms / 1e6}
o_s = (torch.empty((n, 3), dtype=torch.float32, device="cuda"), torch.empty((n,), dtype=torch.float32, device="cuda"),
       torch.empty((n, 3), dtype=torch.float32, device="cuda"))
ms = timed(lambda: gpu.sample(wi, u, material=mid, out=o_s))
res["sample_only"] = {"ms": ms, "Msample_per_s": n / ms / 1e3}
o_p = torch.empty((n,), dtype=torch.float32, device="cuda")
ms = timed(lambda: gpu.pdf(wi, wo, material=mid, out=o_p))
res["pdf_only"] = {"ms": ms, "Mpdf_per_s": n / ms / 1e3, "stream_GBps": 28 * n / ms / 1e6}
ms = timed(lambda: gpu.eval_pdf(wi, wo, material=mid, out=(o_rgb, o_p)))
res["eval_pdf"] = {"ms": ms, "Munits_per_s": n / ms / 1e3, "stream_GBps": 40 * n / ms / 1e6}

# host-pointer path: pageable numpy arrays in, numpy arrays out
m = 16 << 20
hwi, hwo, hu = wi[:m].cpu().numpy(), wo[:m].cpu().numpy(), u[:m].cpu().numpy()
hout = tuple(np.empty(s, np.float32) for s in ((m, 3), (m,), (m, 3), (m,), (m, 3)))
res["host_pointer_eval_sample"] = {}
for threads in (0, 1, 2, 4, 8, 16):
    gpu.set_option(host.OPT_HOST_THREADS, threads)
    gpu.eval_sample(hwi, hwo, hu, material=mid, out=hout)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        gpu.eval_sample(hwi, hwo, hu, material=mid, out=hout)
        best = min(best, time.perf_counter() - t0)
    res["host_pointer_eval_sample"][f"threads_{threads}"] = {
        "units": m, "s": best, "Munits_per_s": m / best / 1e6, "host_GBps": 76 * m / best / 1e9,
        "note": ("staged hipMemcpy path, 4M-unit chunks" if threads == 0 else
                 f"pipelined: {threads} copy thread(s), pinned double buffers of 2^20 units, zero-copy kernel") + " (PCIe-inclusive; never the bench value)"}
gpu.set_option(host.OPT_HOST_THREADS, 4)
# pinned host arrays (mrl_host_alloc): the kernel reads and writes host memory over PCIe itself (zero copy)
import ctypes as C
L, ctx = gpu._lib, gpu._ctx
m = 16 << 20
bufs = {}
for name, floats in (("wi", 3), ("wo", 3), ("u", 2), ("rgb", 3), ("pdf", 1), ("wo2", 3), ("pdf2", 1), ("w", 3)):
    p = C.c_void_p()
    assert L.mrl_host_alloc(ctx, 4 * floats * m, C.byref(p)) == 0
    bufs[name] = p.value
for name, src in (("wi", hwi), ("wo", hwo), ("u", hu)):
    C.memmove(bufs[name], src.ctypes.data, src.nbytes)
gpu.use_own_stream()
def zc():
    rc = L.mrl_eval_sample_batch(ctx, bufs["wi"], bufs["wo"], bufs["u"], None, mid, m, bufs["rgb"], bufs["pdf"], bufs["wo2"], bufs["pdf2"], bufs["w"])
    assert rc == 0
    L.mrl_synchronize(ctx)
zc()
t0 = time.perf_counter()
for _ in range(3):
    zc()
dt = (time.perf_counter() - t0) / 3
got = np.ctypeslib.as_array((C.c_float * (3 * 1000)).from_address(bufs["rgb"])).reshape(-1, 3).copy()
ref = gpu.eval_sample(hwi[:1000], hwo[:1000], hu[:1000], material=mid)[0]
res["pinned_zero_copy_eval_sample"] = {"units": m, "s": dt, "Munits_per_s": m / dt / 1e6, "PCIe_GBps": 76 * m / dt / 1e9,
                                       "matches_staged_path": bool(np.array_equal(got, ref))}
print(json.dumps(res, indent=1))
