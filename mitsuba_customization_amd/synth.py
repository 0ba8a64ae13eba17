"""Synthetic MERL-layout tables and MERL ``.binary`` file helpers (numpy, host side).

No real MERL file ships with the reference (SURVEY.md §8c "MERL data"), so tests and the bench
run on seeded synthetic tables written in the real layout: planar R,G,B, and
inside a plane ``ind = i_pd + n_pd*(i_td + n_td*i_th)``.  Values here are RAW file values,
i.e. before the 1/1500, 1.15/1500, 1.66/1500 channel scales are applied.

Everything is pure integer hashing / closed-form arithmetic (no numpy RNG streams), so a table
is a function of (kind, seed, dims) on every host.
"""
from __future__ import annotations

import numpy as np

MERL_DIMS = (90, 90, 180)
MERL_N = 90 * 90 * 180
MERL_SCALE = (1.0 / 1500.0, 1.15 / 1500.0, 1.66 / 1500.0)
MERL_FILE_BYTES = 12 + 3 * MERL_N * 8  # 34,992,012


def _mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    z = z.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        z += np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _grid(dims):
    n_th, n_td, n_pd = dims
    ith = np.arange(n_th, dtype=np.float64)[:, None, None]
    itd = np.arange(n_td, dtype=np.float64)[None, :, None]
    ipd = np.arange(n_pd, dtype=np.float64)[None, None, :]
    return ith, itd, ipd


def constant_table(value=(300.0, 200.0, 100.0), dims=MERL_DIMS) -> np.ndarray:
    t = np.empty((3,) + tuple(dims), dtype=np.float64)
    for c in range(3):
        t[c] = value[c]
    return t


def affine_table(coef=None, dims=MERL_DIMS) -> np.ndarray:
    """raw[c] = a0 + a1*i_th + a2*i_td + a3*i_pd — trilinear interpolation reproduces it exactly
    away from the clamped ends and the periodic phi_d seam."""
    if coef is None:
        coef = ((50.0, 3.0, 0.5, 0.25), (20.0, 1.0, 2.0, 0.125), (10.0, 0.25, 0.75, 1.5))
    ith, itd, ipd = _grid(dims)
    t = np.empty((3,) + tuple(dims), dtype=np.float64)
    for c in range(3):
        a0, a1, a2, a3 = coef[c]
        t[c] = a0 + a1 * ith + a2 * itd + a3 * ipd
    return t


def onehot_table(index=(10, 20, 30), value=1500.0, dims=MERL_DIMS) -> np.ndarray:
    t = np.zeros((3,) + tuple(dims), dtype=np.float64)
    t[:, index[0], index[1], index[2]] = value
    return t


def noise_table(seed: int, dims=MERL_DIMS, decades: float = 6.0, negative_fraction: float = 0.02) -> np.ndarray:
    """Log-uniform hash noise spanning `decades` decades with a sprinkle of negative
    (below-horizon style) markers; texel-to-texel contrast is O(1) — the worst case for parity."""
    n = int(np.prod(dims))
    idx = np.arange(3 * n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        r = _mix64(idx ^ (np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)))
    u = (r >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    val = 1500.0 * 10.0 ** (-decades / 2 + decades * u) * 1e-2
    r2 = _mix64(r)
    neg = (r2 >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53)) < negative_fraction
    val = np.where(neg, -1.0, val)
    return val.reshape((3,) + tuple(dims))


def ggx_tab_table(seed: int = 0, dims=MERL_DIMS) -> np.ndarray:
    """Analytic GGX + Lambert BRDF tabulated on the MERL grid ('gold-paint-like'): values span
    ~6 decades like measured data; alpha and albedo derive from the seed.  Texel (i,j,k) is the
    BRDF at theta_h=(i/n_th)^2*pi/2, theta_d=j/n_td*pi/2, phi_d=k/n_pd*pi, phi_h=0.
    Below-horizon directions get MERL's negative marker."""
    n_th, n_td, n_pd = dims
    h = _mix64(np.array([seed * 3 + 1, seed * 3 + 2, seed * 3 + 3], dtype=np.uint64))
    uu = (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    alpha = 0.03 + 0.25 * uu[0]
    kd = np.array([0.05 + 0.5 * uu[1], 0.04 + 0.3 * uu[2], 0.02 + 0.2 * uu[0]])
    f0 = np.array([0.9 - 0.3 * uu[2], 0.6 + 0.2 * uu[1], 0.2 + 0.3 * uu[0]])
    ith, itd, ipd = _grid(dims)
    th = (ith / n_th) ** 2 * (np.pi / 2)
    td = itd / n_td * (np.pi / 2)
    pd = ipd / n_pd * np.pi
    # in = R_z(phi_h=0) R_y(theta_h) (sin td cos pd, sin td sin pd, cos td); out = reflect about h
    dx, dy, dz = np.sin(td) * np.cos(pd), np.sin(td) * np.sin(pd), np.cos(td) + 0 * pd
    ct, st = np.cos(th), np.sin(th)
    inx, iny, inz = dx * ct + dz * st, dy + 0 * th, -dx * st + dz * ct
    hx, hz = st, ct
    dot = inx * hx + inz * hz
    outz = 2 * dot * hz - inz
    valid = (inz > 1e-6) & (outz > 1e-6)
    ci, co = np.maximum(inz, 1e-6), np.maximum(outz, 1e-6)
    tan2h = (st / np.maximum(ct, 1e-9)) ** 2
    D = 1.0 / (np.pi * alpha**2 * np.maximum(ct, 1e-9) ** 4 * (1 + tan2h / alpha**2) ** 2)

    def g1(c):
        t2 = (1 - c * c) / (c * c)
        return 2.0 / (1.0 + np.sqrt(1.0 + alpha**2 * t2))

    G = g1(ci) * g1(co)
    spec = D * G / (4 * ci * co)
    t = np.empty((3,) + tuple(dims), dtype=np.float64)
    cd = np.cos(td) + 0 * pd + 0 * th
    for c in range(3):
        F = f0[c] + (1 - f0[c]) * (1 - cd) ** 5
        f = kd[c] / np.pi + F * spec
        t[c] = np.where(valid, f / MERL_SCALE[c], -1.0)
    return t


def ggx_standard_table(seed: int = 0, dims=(32, 32, 64), full: bool = False) -> np.ndarray:
    """The same analytic GGX + Lambert BRDF tabulated in the STANDARD parameterisation (include/merl_hip.h enum mrl_param):
    texel (i,j,k) is the BRDF at theta_i = i/n_0 pi/2, theta_o = j/n_1 pi/2 and azimuth difference k/n_2 pi
    (full = False, mirror-symmetric) or k/n_2 2 pi (full = True; a 1 + 0.3 sin(dphi) factor breaks the mirror symmetry so
    that the two halves of the period differ).  Raw values in MERL file units (divide MERL_SCALE back in on lookup)."""
    n_0, n_1, n_2 = dims
    h = _mix64(np.array([seed * 3 + 1, seed * 3 + 2, seed * 3 + 3], dtype=np.uint64))
    uu = (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    alpha = 0.05 + 0.25 * uu[0]
    kd = np.array([0.05 + 0.5 * uu[1], 0.04 + 0.3 * uu[2], 0.02 + 0.2 * uu[0]])
    f0 = np.array([0.9 - 0.3 * uu[2], 0.6 + 0.2 * uu[1], 0.2 + 0.3 * uu[0]])
    i0, i1, i2 = _grid(dims)
    ti, to = i0 / n_0 * (np.pi / 2), i1 / n_1 * (np.pi / 2)
    dp = i2 / n_2 * (2 * np.pi if full else np.pi)
    wi = (np.sin(ti) + 0 * to + 0 * dp, 0 * ti + 0 * to + 0 * dp, np.cos(ti) + 0 * to + 0 * dp)
    wo = (np.sin(to) * np.cos(dp) + 0 * ti, np.sin(to) * np.sin(dp) + 0 * ti, np.cos(to) + 0 * dp + 0 * ti)
    hx, hy, hz = wi[0] + wo[0], wi[1] + wo[1], wi[2] + wo[2]
    hn = np.sqrt(hx * hx + hy * hy + hz * hz)
    hx, hy, hz = hx / hn, hy / hn, hz / hn
    ct = np.maximum(hz, 1e-9)
    tan2h = (1 - ct * ct) / (ct * ct)
    D = 1.0 / (np.pi * alpha**2 * ct**4 * (1 + tan2h / alpha**2) ** 2)
    ci, co = np.maximum(wi[2], 1e-3), np.maximum(wo[2], 1e-3)

    def g1(c):
        return 2.0 / (1.0 + np.sqrt(1.0 + alpha**2 * (1 - c * c) / (c * c)))

    spec = D * g1(ci) * g1(co) / (4 * ci * co)
    cd = wi[0] * hx + wi[1] * hy + wi[2] * hz
    skew = 1.0 + (0.3 * np.sin(dp) if full else 0.0)
    t = np.empty((3,) + tuple(dims), dtype=np.float64)
    for c in range(3):
        F = f0[c] + (1 - f0[c]) * (1 - cd) ** 5
        t[c] = (kd[c] / np.pi + F * spec) * skew / MERL_SCALE[c]
    return t


def make_table(kind: str, seed: int = 0, dims=MERL_DIMS) -> np.ndarray:
    if kind == "ggx_std":
        return ggx_standard_table(seed, dims=dims, full=False)
    if kind == "ggx_std_full":
        return ggx_standard_table(seed, dims=dims, full=True)
    if kind == "constant":
        return constant_table(dims=dims)
    if kind == "affine":
        return affine_table(dims=dims)
    if kind == "onehot":
        return onehot_table(dims=dims)
    if kind == "noise":
        return noise_table(seed, dims=dims)
    if kind == "ggx_tab":
        return ggx_tab_table(seed, dims=dims)
    raise ValueError(f"unknown table kind {kind!r}")


# ---- MERL .binary file -------------------------------------------------------
def write_merl_binary(path: str, planar: np.ndarray) -> None:
    planar = np.ascontiguousarray(planar, dtype="<f8")
    assert planar.ndim == 4 and planar.shape[0] == 3
    with open(path, "wb") as f:
        np.asarray(planar.shape[1:], dtype="<i4").tofile(f)
        planar.tofile(f)


def read_merl_binary(path: str, require_merl_dims: bool = True) -> np.ndarray:
    with open(path, "rb") as f:
        dims = np.fromfile(f, dtype="<i4", count=3)
        if dims.size != 3 or (dims <= 0).any():
            raise ValueError("bad MERL header")
        n = int(dims[0]) * int(dims[1]) * int(dims[2])
        if require_merl_dims and n != MERL_N:
            raise ValueError(f"dims {tuple(dims)} do not match the MERL grid")
        data = np.fromfile(f, dtype="<f8", count=3 * n)
        if data.size != 3 * n:
            raise ValueError("truncated MERL file")
    return data.reshape((3,) + tuple(int(d) for d in dims))


# ---- n-channel tables (customized_measurement beyond RGB) --------------------------------------
def make_table_nch(kind: str, n_ch: int, seed: int = 0, dims=MERL_DIMS) -> np.ndarray:
    """(n_ch, n_th, n_td, n_pd) raw values.  "noise": independent hash noise per channel; "spectral": the GGX-shaped
    lobe of ggx_tab_table with a smooth per-channel (wavelength-like) albedo and Fresnel ramp; "affine": a different
    affine function of the indices per channel."""
    if kind == "noise":
        planes = [noise_table(seed * 131 + 7 * (c // 3) + 1, dims=dims)[c % 3] for c in range(n_ch)]
        return np.stack(planes, axis=0)
    if kind == "spectral":
        base = ggx_tab_table(seed, dims=dims)                     # 3 planes, raw MERL units (divide the scales back out)
        f = [np.where(base[c] > 0, base[c] * MERL_SCALE[c], -1.0) for c in range(3)]
        out = np.empty((n_ch,) + tuple(dims), dtype=np.float64)
        for c in range(n_ch):
            t = c / max(n_ch - 1, 1)                              # 0 .. 1 across the channels
            w = np.array([(1 - t) ** 2, 2 * t * (1 - t), t * t])  # smooth blend of the three base planes
            mix = w[0] * np.maximum(f[0], 0) + w[1] * np.maximum(f[1], 0) + w[2] * np.maximum(f[2], 0)
            out[c] = np.where(base[0] > 0, mix * (0.8 + 0.4 * np.sin(3.0 * t + seed)), -1.0)
        return out
    if kind == "affine":
        ith, itd, ipd = _grid(dims)
        return np.stack([10.0 + c + (1.0 + 0.5 * c) * ith + (0.5 + 0.25 * c) * itd + (0.125 * (c + 1)) * ipd for c in range(n_ch)], axis=0)
    raise ValueError(f"unknown n-channel table kind {kind!r}")


def write_table_nch(path: str, planar: np.ndarray, dtype="<f8") -> None:
    """customized_measurement file with any channel count: int32 dims[3], then the planes (f64 or f32)."""
    planar = np.ascontiguousarray(planar, dtype=dtype)
    assert planar.ndim == 4
    with open(path, "wb") as f:
        np.asarray(planar.shape[1:], dtype="<i4").tofile(f)
        planar.tofile(f)


# ---- "tensor_file" container (RGL *.bsdf), writer for tests and fixtures -----------------------
_TENSOR_DTYPE = {np.dtype("uint8"): 1, np.dtype("int8"): 2, np.dtype("uint16"): 3, np.dtype("int16"): 4, np.dtype("uint32"): 5,
                 np.dtype("int32"): 6, np.dtype("uint64"): 7, np.dtype("int64"): 8, np.dtype("float16"): 9, np.dtype("float32"): 10,
                 np.dtype("float64"): 11}


def write_tensor_file(path: str, fields: dict) -> None:
    """fields: {name: ndarray}.  Header "tensor_file\\0", version 1.0, field table, payloads (see csrc/merl_tensor_file.hip)."""
    import struct
    names = list(fields)
    arrays = [np.array(fields[k], order="C") for k in names]             # (ascontiguousarray would turn 0-d into 1-d)
    head = 12 + 2 + 4 + sum(2 + len(k.encode()) + 2 + 1 + 8 + 8 * a.ndim for k, a in zip(names, arrays))
    offsets, at = [], head
    for a in arrays:
        at = (at + 7) // 8 * 8                                   # payloads 8-byte aligned
        offsets.append(at)
        at += a.nbytes
    with open(path, "wb") as f:
        f.write(b"tensor_file\0" + struct.pack("<BBI", 1, 0, len(names)))
        for k, a, off in zip(names, arrays, offsets):
            kb = k.encode()
            f.write(struct.pack("<H", len(kb)) + kb + struct.pack("<HBQ", a.ndim, _TENSOR_DTYPE[a.dtype], off))
            f.write(struct.pack(f"<{a.ndim}Q", *a.shape))
        for a, off in zip(arrays, offsets):
            f.write(b"\0" * (off - f.tell()))
            f.write(a.astype(a.dtype.newbyteorder("<")).tobytes())


# ------------------------------------------------------------------ synthetic RGL *.bsdf fields (the adaptive parameterisation)
def make_rgl_fields(seed: int = 0, n_phi: int = 1, n_theta: int = 6, res: int = 12, res_ndf: int = 16, res_sigma: int = 8, reduction: int = 1,
                    n_wavelengths: int = 0) -> dict:
    """Fields of an RGL material-database file with the real names and shapes (what upstream Mitsuba 3's `measured` reads):
    phi_i [n_phi], theta_i [n_theta], ndf [res_ndf, res_ndf], sigma [res_sigma, res_sigma], vndf / luminance
    [n_phi, n_theta, res, res], rgb [n_phi, n_theta, 3, res, res], jacobian [1], description.  n_phi <= 2: isotropic.
    reduction = 2 / 4 (anisotropic only): phi_i covers [-pi, 0] / [-pi, -pi/2], as for a sample with a point symmetry / two mirror planes.
    n_wavelengths > 0: a SPECTRAL file — "spectra" [n_phi, n_theta, n_wavelengths, res, res] over "wavelengths" [n_wavelengths] (ascending,
    unevenly spaced on purpose, 360 - 1000 nm) instead of "rgb".
    No measured file exists offline: the tables are smooth, strictly positive synthetic functions (a lobe + seeded
    low-frequency variation), periodic in every azimuth axis as a measurement is — they exercise every code path of the
    model, they are not a material."""
    rng = np.random.default_rng(seed)

    def smooth(shape, lobe=2.0):
        ny, nx = shape[-2:]
        y, x = np.meshgrid((np.arange(ny) + 0.5) / ny, (np.arange(nx) + 0.5) / nx, indexing="ij")
        out = np.empty(shape, np.float64)
        lead = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        flat = out.reshape(lead, ny, nx)
        for k in range(lead):
            a, b, c, d = rng.uniform(0.5, 3.0, 4)
            px, py = rng.uniform(0.2, 0.8, 2)
            flat[k] = 0.15 + np.exp(-lobe * a * (x - px) ** 2 - lobe * b * (y - py) ** 2) * (1.0 + 0.3 * np.sin(c * 6.0 * x) * np.cos(d * 5.0 * y))
        return out.astype(np.float32)

    theta_i = np.linspace(0.0, 0.5 * np.pi * 0.97, n_theta).astype(np.float32)
    phi_i = (np.zeros(1) if n_phi == 1 else np.linspace(-np.pi, -np.pi + 2 * np.pi / reduction, n_phi)).astype(np.float32)

    def closed(a, slices_too=True):
        # the y axis of every warp is an azimuth (u = (phi + pi) / 2 pi): the rows at u = 0 and u = 1 are the same direction,
        # and so are the phi_i = -pi and phi_i = +pi slices of an anisotropic file — a measured file is periodic there
        a[..., -1, :] = a[..., 0, :]
        if slices_too and n_phi > 1 and reduction == 1:
            a[-1] = a[0]
        return a

    ndf = closed(smooth((res_ndf, res_ndf), 4.0), False)
    sigma = closed((0.4 + smooth((res_sigma, res_sigma), 1.0)).astype(np.float32), False)
    vndf, luminance = closed(smooth((n_phi, n_theta, res, res), 3.0)), closed(smooth((n_phi, n_theta, res, res), 1.0))
    n_values = n_wavelengths if n_wavelengths > 0 else 3
    rgb = closed((0.05 + 0.5 * smooth((n_phi, n_theta, n_values, res, res), 1.5)).astype(np.float32))
    if n_wavelengths > 0:
        steps = rng.uniform(0.5, 1.5, n_wavelengths)
        wl = 360.0 + np.concatenate([[0.0], np.cumsum(steps[:-1])]) * (640.0 / max(float(np.sum(steps[:-1])), 1e-9)) if n_wavelengths > 1 else np.array([550.0])
        return {
            "description": np.frombuffer(b"synthetic spectral RGL-shaped fields (mitsuba_customization_amd.synth.make_rgl_fields)", np.uint8).copy(),
            "phi_i": phi_i, "theta_i": theta_i,
            "ndf": ndf, "sigma": sigma, "vndf": vndf, "luminance": luminance, "spectra": rgb, "wavelengths": wl.astype(np.float32),
            "jacobian": np.array([1], np.uint8),
        }
    return {
        "description": np.frombuffer(b"synthetic RGL-shaped fields (mitsuba_customization_amd.synth.make_rgl_fields)", np.uint8).copy(),
        "phi_i": phi_i, "theta_i": theta_i,
        "ndf": ndf, "sigma": sigma, "vndf": vndf, "luminance": luminance, "rgb": rgb,
        "jacobian": np.array([1], np.uint8),
    }
