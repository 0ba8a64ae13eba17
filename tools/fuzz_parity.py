#!/usr/bin/env python3
"""Parity soak: many seeds x tables x options x entry points, GPU vs the CPU oracle on identical inputs.
Rounds cycle through the whole-array fused call, the wavefront-queue call over a random subset of the slots, and
eval_pdf + sample called separately; through MERL-sized and free-dims (customized_measurement) tables with channel
scales; through both node conventions and disk maps; and (round 4) through the Appendix-B options MRL_OPT_COSINE_FACTOR and
MRL_OPT_NEGATIVE (clamp / keep / renormalise).  Prints the worst relative error per output and the count of values beyond 1e-6
(under KEEP a blend can cancel against the -1 markers: the bound there is relative to the blend's terms, 1e-6 |value| + 1e-6 x the
marker's magnitude [/ pdf for a weight]).   python tools/fuzz_parity.py [rounds] > profiles/r04_fuzz_parity.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mitsuba_customization_amd import host, synth
from oracle import binding as ob          # checker (this is a test tool)

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << 20
worst = {"rgb": 0.0, "weight": 0.0}
beyond = {"rgb": 0, "weight": 0}
exact_fail = 0
total = 0
t0 = time.time()
for r in range(rounds):
    kind = ("ggx_tab", "ggx_tab", "noise")[r % 3]
    node, disk = (r // 3) % 2, (r // 6) % 2
    entry = ("batch", "queue", "eval_pdf+sample")[(r // 2) % 3]
    custom = (r % 4) == 3                                    # customized_measurement: free dims, own channel scales
    cosine, negative = (r // 5) % 2, (r // 7) % 3            # SURVEY.md Appendix B 4 and 2 as options
    rng = np.random.default_rng(r)
    dims = tuple(int(x) for x in (rng.integers(8, 70), rng.integers(8, 70), rng.integers(8, 140))) if custom else synth.MERL_DIMS
    scale = tuple(float(x) for x in rng.uniform(0.2, 3.0, 3) / 1500.0) if custom else synth.MERL_SCALE
    tab = synth.make_table(kind, 1000 + r, dims=dims)
    T = ob.OracleTable(tab, scale=scale)
    wi, wo, u = ob.generate_pairs(0xF00D + r, r * 7919, n)
    ref = ob.eval_sample_multi([T], wi, wo, u, None, ob.make_opts(lookup=1, node=node, disk_map=disk, cosine=cosine, negative=negative))
    with host.MerlHip(0) as g:
        g.set_option(host.OPT_NODE, node); g.set_option(host.OPT_DISK_MAP, disk)
        g.set_option(host.OPT_COSINE_FACTOR, cosine); g.set_option(host.OPT_NEGATIVE, negative)
        mid = g.upload_table(tab, scale=scale) if custom else g.upload_merl(tab)
        d_wi, d_wo, d_u = torch.from_numpy(wi).cuda(), torch.from_numpy(wo).cuda(), torch.from_numpy(u).cuda()
        if entry == "batch":
            got = g.eval_sample(d_wi, d_wo, d_u, material=mid)
        elif entry == "queue":
            keep = torch.rand(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(r)) < 0.37
            queue = keep.nonzero().flatten().to(torch.int32)
            count = torch.tensor([queue.numel()], dtype=torch.int32, device="cuda")
            got = g.eval_sample_queue(d_wi, d_wo, d_u, queue, count, material=mid)
            sel = queue.long().cpu().numpy()
            got = [t[queue.long()] for t in got]
            ref = [x[sel] for x in ref]; wi, wo, u = wi[sel], wo[sel], u[sel]
        else:
            rgb, pdf = g.eval_pdf(d_wi, d_wo, material=mid)
            wo2, pdf2, w = g.sample(d_wi, d_u, material=mid)
            got = (rgb, pdf, wo2, pdf2, w)
        got = [t.cpu().numpy() for t in got]
    for name, k in (("rgb", 0), ("weight", 4)):
        a = got[k].astype(np.float64); b = ref[k].astype(np.float64)
        floor = 1e-30
        if negative == 1:                                    # KEEP: cancellation against the markers
            floor = 1e-6 * max(scale) * (1.0 if k == 0 else 1.0 / np.maximum(ref[3].astype(np.float64), 1e-30)[:, None])
        err = np.abs(a - b) / np.maximum(np.abs(b), 1e-30)
        err = np.where(np.abs(a - b) <= floor, 0.0, err)
        if kind == "noise":       # phi_d ill-conditioned near theta_d -> 0 on the noise table: skip those units
            an = wi / np.linalg.norm(wi, axis=1, keepdims=True); bn = (wo if k == 0 else got[2]) / np.maximum(np.linalg.norm(wo if k == 0 else got[2], axis=1, keepdims=True), 1e-30)
            s = an + bn; e = an - bn
            well = (np.arctan2(np.hypot(s[:, 0], s[:, 1]), s[:, 2]) > 0.02) & (np.arctan2(np.linalg.norm(e, axis=1), np.linalg.norm(s, axis=1)) > 0.02)
            err = err[well]
        worst[name] = max(worst[name], float(err.max()))
        beyond[name] += int((err > 1e-6).sum())
    exact_fail += int((got[1] != ref[1]).sum() + (got[2] != ref[2]).sum() + (got[3] != ref[3]).sum())
    total += len(wi)
print(json.dumps({"units": total, "rounds": rounds, "worst_rel_err": worst, "values_beyond_1e-6": beyond,
                  "bit_mismatches_in_pdf_wo_pdf2": exact_fail, "options_cycled": "node x disk map x cosine factor x negative policy x entry point x table kind",
                  "library": host.build_info(), "seconds": round(time.time() - t0, 1)}))
