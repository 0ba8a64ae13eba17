#!/usr/bin/env python3
"""What the adaptive-parameterisation (RGL) material costs: eval, pdf, sample and the fused eval+sample unit over 16M units,
for a file of the database's isotropic shape (8 theta_i nodes, 32 x 32 warps, 128 x 128 ndf) and an anisotropic one; each with the
search tables in LDS where they fit (the default) and read from memory (MRL_OPT_RGL_SEARCH = 1).   python tools/rgl_rates.py > profiles/r04_rgl_rates.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mitsuba_customization_amd import host, synth

n = 16 << 20
res = {"units": n, "library": host.build_info()}
with host.MerlHip(0) as g:
    g.use_torch_stream()
    wi, wo, u = g.generate_pairs(0x5EED, 0, n)
    for name, shape in (("isotropic_8x32x32", dict(n_phi=1, n_theta=8, res=32, res_ndf=128, res_sigma=64)),
                        ("anisotropic_16x8x32x32", dict(n_phi=16, n_theta=8, res=32, res_ndf=128, res_sigma=64))):
        mid = g.upload_rgl(synth.make_rgl_fields(seed=9, **shape))
        row = {"image_bytes": g.memory_info()["table_bytes"]}
        for search in (0, 1):
            g.set_option(host.OPT_RGL_SEARCH, search)
            sub = {}
            for what, call in (("eval", lambda: g.eval(wi, wo, material=mid)), ("pdf", lambda: g.pdf(wi, wo, material=mid)),
                               ("eval_pdf", lambda: g.eval_pdf(wi, wo, material=mid)),
                               ("sample", lambda: g.sample(wi, u, material=mid)), ("eval_sample", lambda: g.eval_sample(wi, wo, u, material=mid))):
                for _ in range(2):
                    out = call()
                torch.cuda.synchronize()
                g.timer_start()
                for _ in range(5):
                    out = call()
                ms = g.timer_stop() / 5
                sub[what] = {"ms": round(ms, 3), "G_units_per_s": round(n / ms / 1e6, 3)}
            row["search_lds_if_it_fits" if search == 0 else "search_memory"] = sub
        g.set_option(host.OPT_RGL_SEARCH, 0)
        res[name] = row
        g.release_material(mid)
    # a spectral file of the isotropic shape (195 wavelength nodes would be the database's; 32 here), four wavelengths per unit
    mid = g.upload_rgl(synth.make_rgl_fields(seed=9, n_phi=1, n_theta=8, res=32, res_ndf=128, res_sigma=64, n_wavelengths=32))
    wl = torch.rand(n, 4, device="cuda") * 640.0 + 360.0
    row = {"image_bytes": g.memory_info()["table_bytes"], "wavelengths_per_unit": 4}
    for what, call in (("eval", lambda: g.eval_spectral(wi, wo, wl, mid)), ("sample", lambda: g.sample_spectral(wi, u, wl, mid)),
                       ("eval_sample", lambda: g.eval_sample_spectral(wi, wo, u, wl, mid))):
        for _ in range(2):
            out = call()
        torch.cuda.synchronize()
        g.timer_start()
        for _ in range(5):
            out = call()
        ms = g.timer_stop() / 5
        row[what] = {"ms": round(ms, 3), "G_units_per_s": round(n / ms / 1e6, 3)}
    res["spectral_isotropic_8x32x32_32wl"] = row
    g.release_material(mid)
    del wl
    # a batch with material ids: a MERL-sized table, an analytic material and two RGL files, one quarter of the units each
    tab = g.upload_merl(synth.make_table("ggx_tab", 0))
    ggx = g.ggx(0.1, (0.143, 0.375, 1.442), (3.983, 2.386, 1.603))
    r1 = g.upload_rgl(synth.make_rgl_fields(seed=9, n_phi=1, n_theta=8, res=32, res_ndf=128, res_sigma=64))
    r2 = g.upload_rgl(synth.make_rgl_fields(seed=10, n_phi=1, n_theta=8, res=32, res_ndf=128, res_sigma=64))
    ids = torch.tensor([tab, ggx, r1, r2], device="cuda", dtype=torch.int32)
    mat = ids[torch.randint(0, 4, (n,), device="cuda")]
    row = {}
    for what, call in (("eval_sample_mixed", lambda: g.eval_sample(wi, wo, u, mat=mat)),
                       ("eval_sample_table_and_ggx_only", lambda: g.eval_sample(wi, wo, u, mat=ids[:2][torch.randint(0, 2, (n,), device="cuda")]))):
        for _ in range(2):
            out = call()
        torch.cuda.synchronize()
        g.timer_start()
        for _ in range(5):
            out = call()
        ms = g.timer_stop() / 5
        row[what] = {"ms": round(ms, 3), "G_units_per_s": round(n / ms / 1e6, 3)}
    res["mixed_batch_table_ggx_2rgl"] = row
print(json.dumps(res, indent=1))
