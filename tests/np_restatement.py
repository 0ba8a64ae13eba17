"""Second, independent restatement of the path in numpy (vectorised, f64).

It deliberately uses a DIFFERENT formulation from oracle/merl_oracle.c: no Rodrigues rotations,
no acos — the angles come from well-conditioned atan2 forms valid for unit vectors
(s = in+out, e = in-out):
    theta_h = atan2(|s_xy|, s_z)
    theta_d = atan2(|e|, |s|)
    phi_d   = atan2(e_y s_x - e_x s_y, -e_z |s|)      (atan2(in_y, in_x) when s_xy == 0)
This is the formulation the HIP kernels use; agreement with the C oracle (tests/test_oracle_kat.py)
checks both the oracle and the kernels' algebra without either implementation.
"""
import numpy as np

MERL_SCALE = (1.0 / 1500.0, 1.15 / 1500.0, 1.66 / 1500.0)


def unit(v):
    v = np.asarray(v, np.float64)
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def half_diff(in_, out):
    in_, out = unit(in_), unit(out)
    s, e = in_ + out, in_ - out
    ns = np.linalg.norm(s, axis=-1)
    ne = np.linalg.norm(e, axis=-1)
    rho = np.hypot(s[..., 0], s[..., 1])
    th = np.arctan2(rho, s[..., 2])
    td = np.arctan2(ne, ns)
    y = e[..., 1] * s[..., 0] - e[..., 0] * s[..., 1]
    x = -e[..., 2] * ns
    deg = rho == 0.0
    y = np.where(deg, in_[..., 1], y)
    x = np.where(deg, in_[..., 0], x)
    pd = np.arctan2(y, x)
    return th, td, pd


def coords(th, td, pd, dims=(90, 90, 180)):
    n_th, n_td, n_pd = dims
    xh = np.sqrt(np.maximum(th, 0.0) / (np.pi / 2) * n_th * n_th)
    xd = td / (np.pi / 2) * n_td
    pd = np.where(pd < 0, pd + np.pi, pd)
    xp = pd / np.pi * n_pd
    return xh, xd, xp


def scaled_table(planar, scale=MERL_SCALE):
    t = np.asarray(planar, np.float64) * np.asarray(scale, np.float64)[:, None, None, None]
    return np.maximum(t, 0.0)


def lookup(planar, xh, xd, xp, trilinear=True, center=False, scale=MERL_SCALE, phi_clamped=False):
    t = scaled_table(planar, scale)
    _, n_th, n_td, n_pd = t.shape
    if not trilinear:
        ih = np.clip(xh.astype(np.int64), 0, n_th - 1)
        id_ = np.clip(xd.astype(np.int64), 0, n_td - 1)
        ip = np.clip(xp.astype(np.int64), 0, n_pd - 1)
        return t[:, ih, id_, ip].T
    sh = 0.5 if center else 0.0

    def split_c(x, n):
        i = np.clip(np.floor(x).astype(np.int64), 0, n - 1)
        f = np.clip(x - i, 0.0, 1.0)
        return i, np.minimum(i + 1, n - 1), f

    def split_p(x, n):
        fl = np.floor(x)
        i = np.mod(fl.astype(np.int64), n)
        return i, np.mod(i + 1, n), x - fl

    h0, h1, fh = split_c(xh - sh, n_th)
    d0, d1, fd = split_c(xd - sh, n_td)
    p0, p1, fp = split_c(xp - sh, n_pd) if phi_clamped else split_p(xp - sh, n_pd)
    out = 0.0
    for hi, wh in ((h0, 1 - fh), (h1, fh)):
        for di, wd in ((d0, 1 - fd), (d1, fd)):
            for pi, wp in ((p0, 1 - fp), (p1, fp)):
                out = out + (wh * wd * wp)[None, :] * t[:, hi, di, pi]
    return out.T


def eval_merl(planar, wi, wo, trilinear=True, center=False, scale=MERL_SCALE):
    wi = np.asarray(wi, np.float32); wo = np.asarray(wo, np.float32)
    th, td, pd = half_diff(wi.astype(np.float64), wo.astype(np.float64))
    xh, xd, xp = coords(th, td, pd, np.asarray(planar).shape[1:])
    v = lookup(planar, xh, xd, xp, trilinear, center, scale) * wo[:, 2:3].astype(np.float64)
    ok = (wi[:, 2] > 0) & (wo[:, 2] > 0)
    return np.where(ok[:, None], v, 0.0)


def eval_standard(planar, wi, wo, full=False, trilinear=True, center=False, scale=MERL_SCALE):
    """The standard parameterisations (include/merl_hip.h enum mrl_param), formulated differently from the oracle:
    theta = arccos(z) of the unit vector, dphi = phi_o - phi_i from the two azimuths, wrapped."""
    wi = np.asarray(wi, np.float32); wo = np.asarray(wo, np.float32)
    a, b = unit(wi.astype(np.float64)), unit(wo.astype(np.float64))
    n0, n1, n2 = np.asarray(planar).shape[1:]
    ti, to = np.arccos(np.clip(a[:, 2], -1, 1)), np.arccos(np.clip(b[:, 2], -1, 1))
    pi_, po = np.arctan2(a[:, 1], a[:, 0]), np.arctan2(b[:, 1], b[:, 0])
    flat = ((a[:, 0] == 0) & (a[:, 1] == 0)) | ((b[:, 0] == 0) & (b[:, 1] == 0))
    dp = np.where(flat, 0.0, po - pi_)
    dp = np.mod(dp, 2 * np.pi)                                   # [0, 2 pi)
    x0, x1 = ti / (np.pi / 2) * n0, to / (np.pi / 2) * n1
    if full:
        x2 = dp / (2 * np.pi) * n2
    else:
        x2 = np.where(dp > np.pi, 2 * np.pi - dp, dp) / np.pi * n2
    v = lookup(planar, x0, x1, x2, trilinear, center, scale, phi_clamped=not full) * wo[:, 2:3].astype(np.float64)
    ok = (wi[:, 2] > 0) & (wo[:, 2] > 0)
    return np.where(ok[:, None], v, 0.0)


def square_to_cosine_hemisphere(u, mitsuba3=False):
    """Independent formulation of the cosine-hemisphere warp (Shirley & Chiu's concentric map in its textbook form: radius
    and angle, then libm sin / cos in f64) — what a Mitsuba built against libm computes, up to its own Float rounding.  The
    oracle and the kernels instead share a hand-pinned f32 polynomial (oracle/merl_oracle.c sincos_quarter_f32), so this is
    the evidence that the shared code is the right function, not just the same one."""
    u = np.asarray(u, np.float32)
    a = (np.float32(2) * u[:, 0] - np.float32(1)).astype(np.float64)
    b = (np.float32(2) * u[:, 1] - np.float32(1)).astype(np.float64)
    first = ~(np.abs(a) < np.abs(b)) if mitsuba3 else (a * a > b * b)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.where(first, a, b)
        phi = np.where(first, (np.pi / 4) * (b / a), (np.pi / 2) - (np.pi / 4) * (a / b))
    zero = (a == 0) & (b == 0)
    x = np.where(zero, 0.0, r * np.cos(phi)); y = np.where(zero, 0.0, r * np.sin(phi))
    z = np.sqrt(np.maximum(1.0 - x * x - y * y, 0.0))
    return np.stack([x, y, z], axis=1)
