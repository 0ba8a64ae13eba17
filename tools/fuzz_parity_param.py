#!/usr/bin/env python3
"""Parity soak for the table parameterisations (MRL_OPT_TABLE_PARAM): random parameterisation (two thirds of
the rounds one of the standard forms), random dims / scales / channel count (RGB path or 1..32 channels), both node
conventions and disk maps, nearest lookups in a sixth of the rounds, through the whole-array call, the wavefront-queue
call over a random subset and host arrays.  GPU vs the CPU oracle on identical inputs; no unit is excluded for the
standard forms (their angles are cancellation-free on both sides).   python tools/fuzz_parity_param.py [rounds]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mitsuba_customization_amd import host, synth
from oracle import binding as ob          # checker (this is a test tool)

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 48
n = 1 << 18
worst = {"values": 0.0, "weight": 0.0}
beyond = {"values": 0, "weight": 0}
exact_fail, total, flips = 0, 0, 0
seen = {}
t0 = time.time()
for r in range(rounds):
    rng = np.random.default_rng(4000 + r)
    param = (1, 2, 0)[r % 3]
    C = int(rng.choice([3, 3, 3, 1, 2, 4, 6, 16, 32]))
    lookup = 0 if r % 6 == 5 else 1
    node, disk = (r // 3) % 2, (r // 6) % 2
    entry = ("batch", "queue", "host")[(r // 2) % 3]
    kind = "noise" if (param != 0 and r % 2) else ("ggx_std" if param == 1 else "ggx_std_full" if param == 2 else "ggx_tab")
    dims = tuple(int(x) for x in (rng.integers(6, 48), rng.integers(6, 48), rng.integers(6, 96)))
    seen[f"param{param}_C{C}"] = seen.get(f"param{param}_C{C}", 0) + 1
    scale = [float(x) for x in rng.uniform(0.2, 3.0, C)]
    if C == 3:
        tab = synth.make_table(kind, 700 + r, dims)
        scale = [s / 1500.0 for s in scale]
        T = ob.OracleTable(tab, scale, param=param)
        wi, wo, u = ob.generate_pairs(0xFACE + r, r * 15485863, n)
        ref = ob.eval_sample_multi([T], wi, wo, u, None, ob.make_opts(lookup, node, disk))
    else:
        tab = synth.make_table_nch("noise" if kind == "noise" else "spectral", C, 700 + r, dims)
        T = ob.OracleTableNch(tab, scale, param=param)
        wi, wo, u = ob.generate_pairs(0xFACE + r, r * 15485863, n)
        ref = ob.eval_sample_nch([T], wi, wo, u, None, ob.make_opts(lookup, node, disk))
    with host.MerlHip(0) as g:
        g.set_option(host.OPT_LOOKUP, lookup); g.set_option(host.OPT_NODE, node); g.set_option(host.OPT_DISK_MAP, disk)
        g.set_option(host.OPT_TABLE_PARAM, param)
        mid = g.upload_table(tab, scale) if C == 3 else g.upload_table_nch(tab, scale)
        call = (lambda *a, **k: g.eval_sample(*a, **k)) if C == 3 else (lambda *a, **k: g.eval_sample_nch(*a, C, **k))
        if entry == "host":
            got = [np.asarray(t) for t in call(wi, wo, u, material=mid)]
        else:
            d_wi, d_wo, d_u = torch.from_numpy(wi).cuda(), torch.from_numpy(wo).cuda(), torch.from_numpy(u).cuda()
            if entry == "batch":
                got = call(d_wi, d_wo, d_u, material=mid)
            else:
                keep = torch.rand(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(r)) < 0.37
                queue = keep.nonzero().flatten().to(torch.int32)
                count = torch.tensor([queue.numel()], dtype=torch.int32, device="cuda")
                got = (g.eval_sample_queue(d_wi, d_wo, d_u, queue, count, material=mid) if C == 3
                       else g.eval_sample_queue_nch(d_wi, d_wo, d_u, queue, count, C, material=mid))
                sel = queue.long().cpu().numpy()
                got = [t[queue.long()] for t in got]
                ref = [x[sel] for x in ref]; wi, wo, u = wi[sel], wo[sel], u[sel]
            got = [t.cpu().numpy() for t in got]
    for name, k in (("values", 0), ("weight", 4)):
        a = got[k].astype(np.float64); b = ref[k].astype(np.float64)
        err = np.abs(a - b) / np.maximum(np.abs(b), 1e-30)
        err = np.where(np.abs(a - b) <= 1e-30, 0.0, err)
        if param == 0 and kind == "noise":
            continue
        if lookup == 0:                                  # nearest: count units in a neighbouring texel instead
            flips += int((err > 1e-6).any(axis=1).sum())
            continue
        worst[name] = max(worst[name], float(err.max()))
        beyond[name] += int((err > 1e-6).sum())
    exact_fail += int((got[1] != ref[1]).sum() + (got[2] != ref[2]).sum() + (got[3] != ref[3]).sum())
    total += len(wi)
print(json.dumps({"rounds": rounds, "units": total, "cases": seen, "max_rel_err": worst, "values_beyond_1e-6": beyond,
                  "nearest_units_in_a_neighbouring_texel": flips, "bit_mismatches_in_pdf_or_sampled_direction": exact_fail,
                  "seconds": round(time.time() - t0, 1)}, indent=1))
