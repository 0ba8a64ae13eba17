#!/usr/bin/env python3
"""Parity soak for the RGL adaptive-parameterisation material: random synthetic files (isotropic, anisotropic over the whole
azimuth, half- and quarter-azimuth; 1..8 theta_i nodes; 2..24 nodes per warp axis; jacobian flag on / off), whole-array, queue and
host-array calls, GPU against oracle/rgl_oracle.c on identical inputs: eval and pdf directly, the sampled direction against the
oracle's, and the sampled pdf / weight against the oracle evaluated AT the direction the device returned.
    python tools/fuzz_parity_rgl.py [rounds] > profiles/r04_fuzz_parity_rgl.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mitsuba_customization_amd import host, synth
from oracle import binding as ob          # checker (this is a test tool)

def rel(a, b):
    """relative error of every value (north_star's measure; a floor of 1e-30 keeps exact zeros comparable)"""
    b = np.asarray(b, np.float64)
    return np.abs(np.asarray(a, np.float64) - b) / np.maximum(np.abs(b), 1e-30)


def outside_range(B, what, got, wi, wo):
    """is `got` (one unit's values) outside the range the oracle spans over the rounding box of the pair's half vector?"""
    kind = "eval" if what == "eval" else ("pdf" if what in ("pdf", "sample_pdf") else "weight")
    return not B.in_conditioning_range(kind, got, wi, wo)


def spectral_round(r, rng, shape, worst, beyond, n):
    """every fifth round also builds the SPECTRAL file of the same shape (1..24 wavelength nodes) and checks values / weights at four
    per-unit wavelengths (some outside the grid) — counted under "spectral_values" / "spectral_weights" (1e-6 relative, no exemptions:
    random pairs; a near-mirror unit would show up here as a count)"""
    sp = dict(shape); sp["n_wavelengths"] = int(rng.integers(1, 25))
    fields = synth.make_rgl_fields(**sp)
    B = ob.OracleRgl(fields)
    m = n // 4
    wi, wo, u = ob.generate_pairs(0xBEEF + r, r * 104729, m)
    lo, hi = float(fields["wavelengths"][0]), float(fields["wavelengths"][-1])
    wl = rng.uniform(lo - 50.0, hi + 50.0, (m, 4)).astype(np.float32)
    with host.MerlHip(0) as g:
        g.set_option(host.OPT_RGL_SEARCH, r % 2)
        mid = g.upload_rgl(fields)
        val, pdf, wo2, pdf2, w = [np.asarray(t) for t in g.eval_sample_spectral(wi, wo, u, wl, mid)]
    o_val, o_pdf = B.eval_pdf_spectral(wi, wo, wl)
    live = pdf2 > 0
    c_val, c_pdf = B.eval_pdf_spectral(wi[live], wo2[live], wl[live])
    for k, e in (("spectral_values", rel(val, o_val)), ("spectral_weights", rel(w[live], c_val / c_pdf[:, None]))):
        if e.size:
            worst[k] = max(worst.get(k, 0.0), float(e.max()))
            beyond[k] = beyond.get(k, 0) + int(np.count_nonzero(e > 1e-6))
    return m


def soak(rounds, n):
    spectral_units = 0
    worst = {"eval": 0.0, "pdf": 0.0, "sample_pdf": 0.0, "sample_weight": 0.0, "direction_abs": 0.0}
    beyond = {k: 0 for k in worst}
    outside = {k: 0 for k in worst if k != "direction_abs"}
    live_mismatch = 0
    total = 0
    shapes = {}
    worst_case = {}
    t0 = time.time()
    for r in range(rounds):
        rng = np.random.default_rng(7000 + r)
        kind = ("iso", "aniso", "half", "quarter")[r % 4]
        n_phi = {"iso": int(rng.choice([1, 2])), "aniso": int(rng.integers(3, 9)), "half": int(rng.integers(3, 6)), "quarter": int(rng.integers(3, 5))}[kind]
        reduction = {"iso": 1, "aniso": 1, "half": 2, "quarter": 4}[kind]
        shape = dict(seed=7000 + r, n_phi=n_phi, n_theta=int(rng.integers(1, 9)), res=int(rng.integers(2, 25)), res_ndf=int(rng.integers(2, 33)),
                     res_sigma=int(rng.integers(2, 17)), reduction=reduction)
        shapes[kind] = shapes.get(kind, 0) + 1
        fields = synth.make_rgl_fields(**shape)
        fields["jacobian"] = np.array([r % 3 != 2], np.uint8)
        entry = ("batch", "queue", "host")[(r // 4) % 3]
        B = ob.OracleRgl(fields)
        wi, wo, u = ob.generate_pairs(0xFACE + r, r * 7919, n)
        with host.MerlHip(0) as g:
            g.set_option(host.OPT_RGL_SEARCH, (r // 12) % 2)          # the search tables from LDS / from memory
            mid = g.upload_rgl(fields)
            if entry == "host":
                got = [np.asarray(t) for t in g.eval_sample(wi, wo, u, material=mid)]
            else:
                d = [torch.from_numpy(x).cuda() for x in (wi, wo, u)]
                if entry == "batch":
                    got = [t.cpu().numpy() for t in g.eval_sample(*d, material=mid)]
                else:
                    keep = torch.rand(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(r)) < 0.43
                    queue = keep.nonzero().flatten().to(torch.int32)
                    count = torch.tensor([queue.numel()], dtype=torch.int32, device="cuda")
                    out = g.eval_sample_queue(*d, queue, count, material=mid)
                    sel = queue.long().cpu().numpy()
                    got = [t[queue.long()].cpu().numpy() for t in out]
                    wi, wo, u = wi[sel], wo[sel], u[sel]
        if r % 5 == 4:
            spectral_units += spectral_round(r, rng, shape, worst, beyond, n)
        rgb, pdf, wo2, pdf2, w = got
        o_rgb, o_pdf = B.eval_pdf(wi, wo)
        o_wo2, o_pdf2, _ = B.sample(wi, u)
        live = pdf2 > 0
        live_mismatch += int(np.count_nonzero(live != (o_pdf2 > 0)))
        both = live & (o_pdf2 > 0)
        c_rgb, c_pdf = B.eval_pdf(wi[live], wo2[live])
        li = np.nonzero(live)[0]
        errs = {"eval": (rel(rgb, o_rgb), rgb, wi, wo), "pdf": (rel(pdf, o_pdf), pdf, wi, wo), "sample_pdf": (rel(pdf2[live], c_pdf), pdf2[live], wi[li], wo2[li]),
                "sample_weight": (rel(w[live], c_rgb / c_pdf[:, None]), w[live], wi[li], wo2[li]),
                "direction_abs": (np.abs(wo2[both].astype(np.float64) - o_wo2[both]), None, None, None)}
        for k, (e, val, a_in, a_out) in errs.items():
            if e.size:
                if float(e.max()) > worst[k] and k == "sample_pdf":
                    j = int(np.argmax(e.reshape(e.shape[0], -1).max(axis=1)))
                    idx = li[j]
                    d_i = wi[idx].astype(np.float64); d_i /= np.linalg.norm(d_i); d_o = wo2[idx].astype(np.float64); d_o /= np.linalg.norm(d_o)
                    worst_case = {"round": r, "file": shape, "entry": entry, "wi": [float.hex(float(x)) for x in wi[idx]], "wo": [float.hex(float(x)) for x in wo2[idx]],
                                  "gpu_pdf": float(pdf2[idx]), "oracle_pdf_there": float(c_pdf[j]), "half_vector_transverse_length": float(np.linalg.norm((d_i + d_o)[:2]) / np.linalg.norm(d_i + d_o))}
                worst[k] = max(worst[k], float(e.max()))
                over = e > (5e-7 if k == "direction_abs" else 1e-6)
                beyond[k] += int(np.count_nonzero(over))
                if k != "direction_abs" and k in outside:
                    for j in np.nonzero(over.reshape(over.shape[0], -1).any(axis=1))[0]:
                        outside[k] += int(outside_range(B, k, val[j], a_in[j], a_out[j]))
        total += wi.shape[0]
    return {"rounds": rounds, "units": total, "spectral_units": spectral_units, "files": shapes, "worst": {k: float(f"{v:.3g}") for k, v in worst.items()},
            "beyond_1e-6 (direction: 5e-7 absolute)": beyond, "outside_the_oracles_rounding_range": outside,
            "sampled_above_horizon_mismatches": live_mismatch, "worst_sample_pdf_unit": worst_case,
            "error_measure": "|gpu - oracle| / |oracle| for every value (floor 1e-30); sample pdf / weight against the oracle AT the device's direction; "
                             "a value beyond 1e-6 is then held against the range the oracle spans over the rounding box of the pair's half vector (8 f64 ulps, 25 points, widened by a quarter of its width)",
            "library": host.build_info(), "seconds": round(time.time() - t0, 1)}


if __name__ == "__main__":
    print(json.dumps(soak(int(sys.argv[1]) if len(sys.argv) > 1 else 40, 1 << 17), indent=1))
