#!/usr/bin/env python3
"""Why the GGX chi^2 statistic grows with the sample count: the visible-normal sampler of Heitz & d'Eon 2014 (the
oracle's = upstream's = the kernel's) inverts its CDF through a rational fit; an exact sampler (Heitz 2018) passes against
the same pdf.  CPU only (the oracle is the checker here).   python tools/chi2_diag.py"""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as ob
from scipy import stats
alpha = float(np.float32(0.1)); eta=[0.143,0.375,1.442]; k=[3.983,2.386,1.603]
G = ob.OracleGgx(alpha, eta, k)
wi_dir = np.array([0.5, 0.3, 0.8124], np.float32)
n = 1 << 22
rng = np.random.default_rng(1)
u = rng.random((n, 2)).astype(np.float32)
wi = np.repeat(wi_dir[None, :], n, 0)
def chi2(wo, ok, nz=32, nphi=64, sub=16):
    z = np.clip(wo[:, 2], 0, 1 - 1e-7); phi = np.arctan2(wo[:, 1], wo[:, 0])
    iz = np.clip((z * nz).astype(int), 0, nz - 1); ip = np.clip(((phi + np.pi) / (2 * np.pi) * nphi).astype(int), 0, nphi - 1)
    counts = np.bincount((iz * nphi + ip)[ok], minlength=nz * nphi).astype(float)
    zs = (np.arange(nz * sub) + 0.5) / (nz * sub); ps = (np.arange(nphi * sub) + 0.5) / (nphi * sub) * 2 * np.pi - np.pi
    Z, P = np.meshgrid(zs, ps, indexing="ij"); r = np.sqrt(1 - Z * Z)
    q = np.stack([r * np.cos(P), r * np.sin(P), Z], -1).reshape(-1, 3).astype(np.float32)
    dens = G.pdf(np.repeat(wi_dir[None, :], q.shape[0], 0), q).astype(np.float64).reshape(nz, sub, nphi, sub)
    expected = (dens.mean(axis=(1, 3)) / nz * (2 * np.pi / nphi) * n).reshape(-1)
    dense = expected >= 5
    obs = np.concatenate([counts[dense], [counts[~dense].sum(), (~ok).sum()]])
    exp = np.concatenate([expected[dense], [expected[~dense].sum(), max(n - expected.sum(), 0)]])
    keep = exp > 0
    stat = (((obs - exp) ** 2) / np.where(keep, exp, 1))[keep].sum()
    return round(stat, 1), int(keep.sum()) - 1, stats.chi2.sf(stat, int(keep.sum()) - 1)
wo, pdf, w = G.sample(wi, u)
print("Heitz-d'Eon 2014 sampler (oracle):", chi2(wo, pdf > 0))
# exact sampler: Heitz 2018 (Sampling the GGX Distribution of Visible Normals), f64
V = wi_dir.astype(np.float64); V /= np.linalg.norm(V)
Vh = np.array([alpha * V[0], alpha * V[1], V[2]]); Vh /= np.linalg.norm(Vh)
lensq = Vh[0] ** 2 + Vh[1] ** 2
T1 = np.array([-Vh[1], Vh[0], 0]) / np.sqrt(lensq); T2 = np.cross(Vh, T1)
U = rng.random((n, 2))
r = np.sqrt(U[:, 0]); ph = 2 * np.pi * U[:, 1]
t1 = r * np.cos(ph); t2 = r * np.sin(ph); s = 0.5 * (1 + Vh[2])
t2 = (1 - s) * np.sqrt(1 - t1 ** 2) + s * t2
Nh = t1[:, None] * T1 + t2[:, None] * T2 + np.sqrt(np.maximum(0, 1 - t1 ** 2 - t2 ** 2))[:, None] * Vh
Ne = np.stack([alpha * Nh[:, 0], alpha * Nh[:, 1], np.maximum(0, Nh[:, 2])], -1); Ne /= np.linalg.norm(Ne, axis=1, keepdims=True)
c = Ne @ V
wo2 = 2 * c[:, None] * Ne - V
print("Heitz 2018 exact sampler       :", chi2(wo2.astype(np.float32), wo2[:, 2] > 0))
