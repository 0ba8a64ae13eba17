"""Diagnostic behind tests/test_gpu_parity.py::test_adversarial_directions_match_oracle: on ill-conditioned direction
pairs (theta_h or theta_d < 0.02 rad) of the noise table, how many ulps of rounding in the oracle's acos arguments
(c in _conditioning_range) are needed before the oracle's lookup range contains the device's value."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_parity as t
from oracle import binding as ob
from mitsuba_customization_amd import host, synth
import torch

tab = synth.make_table("noise", 5)
T = ob.OracleTable(tab)
rng = np.random.default_rng(2024)
wi, wo = t._adversarial_pairs(rng, 60000)
a = t._unit(wi.astype(np.float64)); b = t._unit(wo.astype(np.float64))
s = a + b; e = a - b
th = np.arctan2(np.hypot(s[:, 0], s[:, 1]), s[:, 2]); td = np.arctan2(np.linalg.norm(e, axis=1), np.linalg.norm(s, axis=1))
want = T.eval(wi, wo).astype(np.float64)
cos_o = wo[:, 2].astype(np.float64)
with host.MerlHip(0) as g:
    mid = g.upload_merl(tab)
    for variant in (0, 3):
        g.set_option(host.OPT_KERNEL, variant)
        got = g.eval(torch.from_numpy(wi).cuda(), torch.from_numpy(wo).cuda(), material=mid).cpu().numpy().astype(np.float64)
        ill = np.nonzero(~((th > 0.02) & (td > 0.02)))[0]
        rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
        left = ill[(rel[ill] > 1e-6).any(axis=1)]
        print(f"variant {variant}: {len(ill)} ill units, {len(left)} beyond 1e-6 of the oracle; max rel {rel[ill].max():.3g}", flush=True)
        for c in (1, 2, 4, 8, 16, 64):
            lo, hi = t._conditioning_range(T, a[left], b[left], cos_o[left], th[left], td[left], c=float(c))
            inside = ((got[left] >= lo * (1 - 1e-6) - 1e-30) & (got[left] <= hi * (1 + 1e-6) + 1e-30)).all(axis=1)
            print(f"  c = {c}: {inside.sum()} contained, {len(left) - inside.sum()} left", flush=True)
            left = left[~inside]
            if not len(left):
                break
        for i in left[:8]:
            print("   left:", i, "th", th[i], "td", td[i], "got", got[i], "want", want[i], "wi", wi[i], "wo", wo[i], ob.half_diff(a[i], b[i]))
