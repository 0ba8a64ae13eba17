"""ctypes binding of the CPU oracle (oracle/libmerl_oracle.so).

TEST INFRASTRUCTURE: import this only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg (see merl_oracle.h).  PARITY UNPINNED — the reference ships no source.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmerl_oracle.so")

LOOKUP_NEAREST, LOOKUP_TRILINEAR = 0, 1
NODE_INTEGER, NODE_CENTER = 0, 1
DISK_MITSUBA06, DISK_MITSUBA3 = 0, 1
PARAM_HALF_DIFF, PARAM_STANDARD, PARAM_STANDARD_FULL = 0, 1, 2
COSINE_INCLUDED, COSINE_OMITTED = 0, 1
NEGATIVE_CLAMP, NEGATIVE_KEEP, NEGATIVE_RENORMALISE = 0, 1, 2


class Opts(C.Structure):
    _fields_ = [("lookup", C.c_int), ("node", C.c_int), ("disk_map", C.c_int), ("cosine", C.c_int), ("negative", C.c_int)]


class Table(C.Structure):
    _fields_ = [("n_th", C.c_int), ("n_td", C.c_int), ("n_pd", C.c_int),
                ("data", C.POINTER(C.c_double)), ("scale", C.c_double * 3), ("param", C.c_int)]


class Sampling(C.Structure):
    _fields_ = [("n", C.c_int), ("s", C.POINTER(C.c_double)), ("cdf", C.POINTER(C.c_double)), ("c", C.POINTER(C.c_double)), ("alpha", C.c_double)]


class Sampling2d(C.Structure):
    _fields_ = [("n_i", C.c_int), ("rows", C.POINTER(Sampling))]


class TableNch(C.Structure):
    _fields_ = [("n_th", C.c_int), ("n_td", C.c_int), ("n_pd", C.c_int), ("n_ch", C.c_int),
                ("data", C.POINTER(C.c_double)), ("scale", C.POINTER(C.c_double)), ("param", C.c_int)]


class Ggx(C.Structure):
    _fields_ = [("alpha", C.c_double), ("eta", C.c_double * 3), ("k", C.c_double * 3)]


class Bsdf(C.Structure):
    _fields_ = [("vtbl", C.c_void_p), ("table", Table), ("ggx", Ggx), ("opts", Opts)]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("merl_oracle.c", "merl_oracle.h", "rgl_oracle.c", "rgl_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.orc_read_table.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_int)]
        L.orc_write_table.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_half_diff.argtypes = [C.POINTER(C.c_double)] * 2 + [C.POINTER(C.c_double)] * 4
        L.orc_standard_angles.argtypes = [C.POINTER(C.c_double)] * 2 + [C.POINTER(C.c_double)] * 3
        for nm in ("orc_theta_half_index", "orc_theta_diff_index", "orc_phi_diff_index"):
            getattr(L, nm).argtypes = [C.POINTER(Table), C.c_double]
            getattr(L, nm).restype = C.c_int
        L.orc_coords.argtypes = [C.POINTER(Table)] + [C.c_double] * 3 + [C.POINTER(C.c_double)] * 3
        L.orc_lookup.argtypes = [C.POINTER(Table), C.POINTER(Opts)] + [C.c_double] * 3 + [C.POINTER(C.c_double)]
        L.orc_eval_batch.argtypes = [C.POINTER(Table), C.POINTER(Opts), fp, fp, C.c_size_t, fp]
        L.orc_pdf_batch.argtypes = [fp, fp, C.c_size_t, fp]
        L.orc_sample_batch.argtypes = [C.POINTER(Table), C.POINTER(Opts), fp, fp, C.c_size_t, fp, fp, fp]
        L.orc_eval_sample_batch_multi.argtypes = [C.POINTER(Table), C.c_int, C.POINTER(Opts), fp, fp, fp,
                                                  C.POINTER(C.c_int32), C.c_size_t, fp, fp, fp, fp, fp]
        L.orc_square_to_cosine_hemisphere.argtypes = [C.c_int, fp, fp]
        L.orc_build_sampling.argtypes = [C.POINTER(Table), C.POINTER(Sampling)]
        L.orc_free_sampling.argtypes = [C.POINTER(Sampling)]
        L.orc_pdf_table_batch.argtypes = [C.POINTER(Sampling), fp, fp, C.c_size_t, fp]
        L.orc_sample_table_batch.argtypes = [C.POINTER(Table), C.POINTER(Opts), C.POINTER(Sampling), fp, fp, C.c_size_t, fp, fp, fp]
        L.orc_build_sampling2d.argtypes = [C.POINTER(Table), C.POINTER(Opts), C.c_int, C.POINTER(Sampling2d)]
        L.orc_sampling2d_from_arrays.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(Sampling2d)]
        L.orc_free_sampling2d.argtypes = [C.POINTER(Sampling2d)]
        L.orc_pdf_table2d_batch.argtypes = [C.POINTER(Sampling2d), fp, fp, C.c_size_t, fp]
        L.orc_sample_table2d_batch.argtypes = [C.POINTER(Table), C.POINTER(Opts), C.POINTER(Sampling2d), fp, fp, C.c_size_t, fp, fp, fp]
        L.orc_build_sampling_nch.argtypes = [C.POINTER(TableNch), C.POINTER(Sampling)]
        L.orc_eval_sample_batch_nch.argtypes = [C.POINTER(TableNch), C.c_int, C.c_int, C.POINTER(Opts), C.POINTER(Sampling), fp, fp, fp,
                                                C.POINTER(C.c_int32), C.c_size_t, fp, fp, fp, fp, fp]
        L.orc_lookup_nch.argtypes = [C.POINTER(TableNch), C.POINTER(Opts)] + [C.c_double] * 3 + [C.POINTER(C.c_double)]
        L.orc_ggx_eval_batch.argtypes = [C.POINTER(Ggx), fp, fp, C.c_size_t, fp]
        L.orc_ggx_pdf_batch.argtypes = [C.POINTER(Ggx), fp, fp, C.c_size_t, fp]
        L.orc_ggx_sample_batch.argtypes = [C.POINTER(Ggx), fp, fp, C.c_size_t, fp, fp, fp]
        L.orc_generate_pairs.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, fp, fp, fp]
        L.orc_generate_materials.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_int, C.POINTER(C.c_int32)]
        L.orc_bsdf_init_merl.argtypes = [C.POINTER(Bsdf), C.POINTER(C.c_double), C.POINTER(Opts)]
        L.orc_bsdf_init_ggx.argtypes = [C.POINTER(Bsdf), C.POINTER(Ggx)]
        for nm in ("orc_bench_eval_sample", "orc_bench_eval"):
            getattr(L, nm).argtypes = [C.POINTER(Bsdf), C.c_uint64, C.c_uint64, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
            getattr(L, nm).restype = C.c_double
        _lib = L
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def make_opts(lookup=LOOKUP_TRILINEAR, node=NODE_INTEGER, disk_map=DISK_MITSUBA06, cosine=COSINE_INCLUDED, negative=NEGATIVE_CLAMP) -> Opts:
    return Opts(lookup, node, disk_map, cosine, negative)


class OracleTable:
    """A planar f64 table (3, n_th, n_td, n_pd) + channel scales, as the oracle sees it."""

    def __init__(self, planar: np.ndarray, scale=None, param=PARAM_HALF_DIFF):
        self.planar = np.ascontiguousarray(planar, dtype=np.float64)
        assert self.planar.ndim == 4 and self.planar.shape[0] == 3
        if scale is None:
            scale = (1.0 / 1500.0, 1.15 / 1500.0, 1.66 / 1500.0)
        self.param = int(param)
        self.c = Table(self.planar.shape[1], self.planar.shape[2], self.planar.shape[3],
                       _dp(self.planar), (C.c_double * 3)(*scale), self.param)

    def eval(self, wi, wo, opts=None):
        opts = opts or make_opts()
        wi, pwi = _f32(wi); wo, pwo = _f32(wo)
        n = wi.shape[0]
        out = np.empty((n, 3), np.float32)
        lib().orc_eval_batch(C.byref(self.c), C.byref(opts), pwi, pwo, n, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def sample(self, wi, u, opts=None):
        opts = opts or make_opts()
        wi, pwi = _f32(wi); u, pu = _f32(u)
        n = wi.shape[0]
        wo = np.empty((n, 3), np.float32); pdf = np.empty(n, np.float32); w = np.empty((n, 3), np.float32)
        fp = C.POINTER(C.c_float)
        lib().orc_sample_batch(C.byref(self.c), C.byref(opts), pwi, pu, n,
                               wo.ctypes.data_as(fp), pdf.ctypes.data_as(fp), w.ctypes.data_as(fp))
        return wo, pdf, w

    # ---- table importance sampling ----
    def sampling(self):
        if getattr(self, "_sampling", None) is None:
            self._sampling = Sampling()
            assert lib().orc_build_sampling(C.byref(self.c), C.byref(self._sampling)) == 0
        return self._sampling

    def sampling_arrays(self):
        sp = self.sampling()
        n = sp.n
        return (np.ctypeslib.as_array(sp.s, (n + 1,)).copy(), np.ctypeslib.as_array(sp.cdf, (n + 1,)).copy(),
                np.ctypeslib.as_array(sp.c, (n,)).copy())

    def pdf_table(self, wi, wo):
        wi, pwi = _f32(wi); wo, pwo = _f32(wo)
        out = np.empty(wi.shape[0], np.float32)
        lib().orc_pdf_table_batch(C.byref(self.sampling()), pwi, pwo, wi.shape[0], out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def sample_table(self, wi, u, opts=None):
        opts = opts or make_opts()
        wi, pwi = _f32(wi); u, pu = _f32(u)
        n = wi.shape[0]
        fp = C.POINTER(C.c_float)
        wo = np.empty((n, 3), np.float32); pdf = np.empty(n, np.float32); w = np.empty((n, 3), np.float32)
        lib().orc_sample_table_batch(C.byref(self.c), C.byref(opts), C.byref(self.sampling()), pwi, pu, n,
                                     wo.ctypes.data_as(fp), pdf.ctypes.data_as(fp), w.ctypes.data_as(fp))
        return wo, pdf, w

    # ---- P(theta_h | theta_i): the two-dimensional conditional sampler (merl_oracle.h) ----
    def sampling2d(self, n_i=32, opts=None, flat=None):
        """The oracle's own build, or — flat given: [n_i, 2 n_th + 1] doubles — a table built elsewhere (the device's)."""
        opts = opts or make_opts()
        sp = Sampling2d()
        if flat is None:
            assert lib().orc_build_sampling2d(C.byref(self.c), C.byref(opts), int(n_i), C.byref(sp)) == 0
        else:
            flat = np.ascontiguousarray(flat, dtype=np.float64)
            s = np.ascontiguousarray(self.sampling_arrays()[0])
            assert flat.shape == (n_i, 2 * self.c.n_th + 1)
            assert lib().orc_sampling2d_from_arrays(int(n_i), self.c.n_th, _dp(s), _dp(flat), C.byref(sp)) == 0
        return sp

    def sampling2d_arrays(self, sp):
        n = self.c.n_th
        return np.stack([np.concatenate([np.ctypeslib.as_array(sp.rows[i].cdf, (n + 1,)), np.ctypeslib.as_array(sp.rows[i].c, (n,))])
                         for i in range(sp.n_i)]).copy()

    def pdf_table2d(self, sp, wi, wo):
        wi, pwi = _f32(wi); wo, pwo = _f32(wo)
        out = np.empty(wi.shape[0], np.float32)
        lib().orc_pdf_table2d_batch(C.byref(sp), pwi, pwo, wi.shape[0], out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def sample_table2d(self, sp, wi, u, opts=None):
        opts = opts or make_opts()
        wi, pwi = _f32(wi); u, pu = _f32(u)
        n = wi.shape[0]
        fp = C.POINTER(C.c_float)
        wo = np.empty((n, 3), np.float32); pdf = np.empty(n, np.float32); w = np.empty((n, 3), np.float32)
        lib().orc_sample_table2d_batch(C.byref(self.c), C.byref(opts), C.byref(sp), pwi, pu, n,
                                       wo.ctypes.data_as(fp), pdf.ctypes.data_as(fp), w.ctypes.data_as(fp))
        return wo, pdf, w

    def lookup(self, th, td, pd, opts=None):
        opts = opts or make_opts()
        out = (C.c_double * 3)()
        lib().orc_lookup(C.byref(self.c), C.byref(opts), th, td, pd, out)
        return np.array(out[:])

    def coords(self, th, td, pd):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        lib().orc_coords(C.byref(self.c), th, td, pd, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value


class OracleTableNch:
    """A planar f64 table (n_ch, n_th, n_td, n_pd) + per-channel scales, as the oracle sees it."""

    def __init__(self, planar: np.ndarray, scale=None, param=PARAM_HALF_DIFF):
        self.planar = np.ascontiguousarray(planar, dtype=np.float64)
        assert self.planar.ndim == 4
        self.param = int(param)
        self.n_ch = int(self.planar.shape[0])
        self.scale = np.ascontiguousarray(np.ones(self.n_ch) if scale is None else scale, dtype=np.float64)
        assert self.scale.shape == (self.n_ch,)
        self.c = TableNch(self.planar.shape[1], self.planar.shape[2], self.planar.shape[3], self.n_ch, _dp(self.planar), _dp(self.scale), self.param)
        self._sampling = None

    def sampling(self):
        if self._sampling is None:
            self._sampling = Sampling()
            assert lib().orc_build_sampling_nch(C.byref(self.c), C.byref(self._sampling)) == 0
        return self._sampling

    def lookup(self, th, td, pd, opts=None):
        opts = opts or make_opts()
        out = (C.c_double * self.n_ch)()
        lib().orc_lookup_nch(C.byref(self.c), C.byref(opts), th, td, pd, out)
        return np.array(out[:])


def eval_sample_nch(tables, wi, wo, u, mat=None, opts=None, table_sampling=False, n_ch=None):
    """tables: list[OracleTableNch]; returns values[n, C], pdf, wo2, pdf2, weight[n, C]."""
    opts = opts or make_opts()
    n_ch = n_ch or tables[0].n_ch
    arr = (TableNch * len(tables))(*[t.c for t in tables])
    sp = None
    if table_sampling:
        sp = (Sampling * len(tables))(*[t.sampling() for t in tables])
    wi, pwi = _f32(wi); wo, pwo = _f32(wo); u, pu = _f32(u)
    n = wi.shape[0]
    pm = None
    if mat is not None:
        mat = np.ascontiguousarray(mat, dtype=np.int32)
        pm = mat.ctypes.data_as(C.POINTER(C.c_int32))
    fp = C.POINTER(C.c_float)
    val = np.empty((n, n_ch), np.float32); p = np.empty(n, np.float32)
    wo2 = np.empty((n, 3), np.float32); p2 = np.empty(n, np.float32); w = np.empty((n, n_ch), np.float32)
    lib().orc_eval_sample_batch_nch(arr, len(tables), n_ch, C.byref(opts), sp, pwi, pwo, pu, pm, n,
                                    val.ctypes.data_as(fp), p.ctypes.data_as(fp), wo2.ctypes.data_as(fp),
                                    p2.ctypes.data_as(fp), w.ctypes.data_as(fp))
    return val, p, wo2, p2, w


def pdf(wi, wo):
    wi, pwi = _f32(wi); wo, pwo = _f32(wo)
    out = np.empty(wi.shape[0], np.float32)
    lib().orc_pdf_batch(pwi, pwo, wi.shape[0], out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def eval_sample_multi(tables, wi, wo, u, mat, opts=None):
    """tables: list[OracleTable]; returns rgb, pdf, wo2, pdf2, weight."""
    opts = opts or make_opts()
    arr = (Table * len(tables))(*[t.c for t in tables])
    wi, pwi = _f32(wi); wo, pwo = _f32(wo); u, pu = _f32(u)
    n = wi.shape[0]
    pm = None
    if mat is not None:
        mat = np.ascontiguousarray(mat, dtype=np.int32)
        pm = mat.ctypes.data_as(C.POINTER(C.c_int32))
    fp = C.POINTER(C.c_float)
    rgb = np.empty((n, 3), np.float32); p = np.empty(n, np.float32)
    wo2 = np.empty((n, 3), np.float32); p2 = np.empty(n, np.float32); w = np.empty((n, 3), np.float32)
    lib().orc_eval_sample_batch_multi(arr, len(tables), C.byref(opts), pwi, pwo, pu, pm, n,
                                      rgb.ctypes.data_as(fp), p.ctypes.data_as(fp), wo2.ctypes.data_as(fp),
                                      p2.ctypes.data_as(fp), w.ctypes.data_as(fp))
    return rgb, p, wo2, p2, w


def standard_angles(in_vec, out_vec):
    a = np.ascontiguousarray(in_vec, np.float64); b = np.ascontiguousarray(out_vec, np.float64)
    r = [C.c_double() for _ in range(3)]
    lib().orc_standard_angles(_dp(a), _dp(b), *[C.byref(x) for x in r])
    return tuple(x.value for x in r)  # theta_i, theta_o, dphi


def half_diff(in_vec, out_vec):
    a = np.ascontiguousarray(in_vec, np.float64); b = np.ascontiguousarray(out_vec, np.float64)
    r = [C.c_double() for _ in range(4)]
    lib().orc_half_diff(_dp(a), _dp(b), *[C.byref(x) for x in r])
    return tuple(x.value for x in r)  # theta_half, phi_half, theta_diff, phi_diff


def square_to_cosine_hemisphere(u, disk_map=DISK_MITSUBA06):
    u = np.ascontiguousarray(u, np.float32).reshape(-1, 2)
    out = np.empty((u.shape[0], 3), np.float32)
    fp = C.POINTER(C.c_float)
    for i in range(u.shape[0]):
        lib().orc_square_to_cosine_hemisphere(disk_map, u[i].ctypes.data_as(fp), out[i].ctypes.data_as(fp))
    return out


class OracleGgx:
    def __init__(self, alpha, eta, k):
        self.c = Ggx(alpha, (C.c_double * 3)(*eta), (C.c_double * 3)(*k))

    def eval(self, wi, wo):
        wi, pwi = _f32(wi); wo, pwo = _f32(wo)
        out = np.empty((wi.shape[0], 3), np.float32)
        lib().orc_ggx_eval_batch(C.byref(self.c), pwi, pwo, wi.shape[0], out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def pdf(self, wi, wo):
        wi, pwi = _f32(wi); wo, pwo = _f32(wo)
        out = np.empty(wi.shape[0], np.float32)
        lib().orc_ggx_pdf_batch(C.byref(self.c), pwi, pwo, wi.shape[0], out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def sample(self, wi, u):
        wi, pwi = _f32(wi); u, pu = _f32(u)
        n = wi.shape[0]
        fp = C.POINTER(C.c_float)
        wo = np.empty((n, 3), np.float32); pdf_ = np.empty(n, np.float32); w = np.empty((n, 3), np.float32)
        lib().orc_ggx_sample_batch(C.byref(self.c), pwi, pu, n, wo.ctypes.data_as(fp), pdf_.ctypes.data_as(fp), w.ctypes.data_as(fp))
        return wo, pdf_, w


def generate_pairs(seed, first, n):
    fp = C.POINTER(C.c_float)
    wi = np.empty((n, 3), np.float32); wo = np.empty((n, 3), np.float32); u = np.empty((n, 2), np.float32)
    lib().orc_generate_pairs(seed, first, n, wi.ctypes.data_as(fp), wo.ctypes.data_as(fp), u.ctypes.data_as(fp))
    return wi, wo, u


def generate_materials(seed, first, n, n_materials):
    mat = np.empty(n, np.int32)
    lib().orc_generate_materials(seed, first, n, n_materials, mat.ctypes.data_as(C.POINTER(C.c_int32)))
    return mat


def bench_merl(planar, n, n_threads, seed=0x5EED, with_sample=True, opts=None):
    """Times n units through the Mitsuba-0.6-style virtual bsdf; returns (seconds, checksum)."""
    opts = opts or make_opts()
    planar = np.ascontiguousarray(planar, np.float64)
    b = Bsdf()
    lib().orc_bsdf_init_merl(C.byref(b), _dp(planar), C.byref(opts))
    chk = C.c_double()
    fn = lib().orc_bench_eval_sample if with_sample else lib().orc_bench_eval
    s = fn(C.byref(b), seed, 0, n, n_threads, C.byref(chk))
    return s, chk.value


def bench_ggx(alpha, eta, k, n, n_threads, seed=0x5EED, with_sample=True):
    g = Ggx(alpha, (C.c_double * 3)(*eta), (C.c_double * 3)(*k))
    b = Bsdf()
    lib().orc_bsdf_init_ggx(C.byref(b), C.byref(g))
    chk = C.c_double()
    fn = lib().orc_bench_eval_sample if with_sample else lib().orc_bench_eval
    s = fn(C.byref(b), seed, 0, n, n_threads, C.byref(chk))
    return s, chk.value


# ------------------------------------------------------------------ the RGL adaptive-parameterisation BSDF (rgl_oracle.c)
class RglWarp(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("n_dim", C.c_int), ("n_par", C.c_int * 3), ("stride", C.c_int * 3), ("n_slices", C.c_int),
                ("par", C.POINTER(C.c_float) * 3), ("data", C.POINTER(C.c_float)), ("marg", C.POINTER(C.c_float)), ("cond", C.POINTER(C.c_float)),
                ("normalized", C.c_int)]


class RglBsdf(C.Structure):
    _fields_ = [("isotropic", C.c_int), ("jacobian", C.c_int), ("reduction", C.c_int), ("n_wavelengths", C.c_int), ("ndf", RglWarp), ("sigma", RglWarp), ("vndf", RglWarp), ("luminance", RglWarp), ("rgb", RglWarp)]


def _rgl_lib():
    L = lib()
    if not getattr(L, "_rgl_ready", False):
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        L.rgl_warp_init.argtypes = [C.POINTER(RglWarp), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(fp), fp, C.c_int, C.c_int]
        L.rgl_warp_free.argtypes = [C.POINTER(RglWarp)]
        L.rgl_warp_eval.argtypes = [C.POINTER(RglWarp), dp, dp]; L.rgl_warp_eval.restype = C.c_double
        L.rgl_warp_sample.argtypes = [C.POINTER(RglWarp), dp, dp, dp, dp]
        L.rgl_warp_invert.argtypes = [C.POINTER(RglWarp), dp, dp, dp, dp]
        L.rgl_bsdf_init.argtypes = [C.POINTER(RglBsdf), C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, fp, fp, C.c_int]
        L.rgl_bsdf_free.argtypes = [C.POINTER(RglBsdf)]
        L.rgl_eval_pdf_batch.argtypes = [C.POINTER(RglBsdf), fp, fp, C.c_size_t, fp, fp]
        L.rgl_sample_batch.argtypes = [C.POINTER(RglBsdf), fp, fp, C.c_size_t, fp, fp, fp]
        L.rgl_bsdf_init_spectral.argtypes = [C.POINTER(RglBsdf), C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, fp, C.c_int, fp, fp, C.c_int]
        L.rgl_eval_pdf_spectral_batch.argtypes = [C.POINTER(RglBsdf), fp, fp, fp, C.c_int, C.c_size_t, fp, fp]
        L.rgl_sample_spectral_batch.argtypes = [C.POINTER(RglBsdf), fp, fp, fp, C.c_int, C.c_size_t, fp, fp, fp]
        L.rgl_half_vector.argtypes = [C.POINTER(RglBsdf), fp, fp, dp, dp]; L.rgl_half_vector.restype = C.c_int
        L.rgl_eval_pdf_half.argtypes = [C.POINTER(RglBsdf), dp, dp, fp, fp]
        L._rgl_ready = True
    return L


class OracleWarp:
    """rgl_warp: data [n_slices..., ny, nx] float32, params: list of ascending float32 grids (one per leading axis)."""

    def __init__(self, data, params=(), normalize=True, build_cdf=True):
        L = _rgl_lib()
        self.data = np.ascontiguousarray(data, np.float32)
        self.params = [np.ascontiguousarray(p, np.float32) for p in params]
        assert self.data.ndim == 2 + len(self.params)
        ny, nx = self.data.shape[-2:]
        n_par = (C.c_int * 3)(*[len(p) for p in self.params] + [0] * (3 - len(self.params)))
        fp = C.POINTER(C.c_float)
        par = (fp * 3)(*[p.ctypes.data_as(fp) for p in self.params] + [None] * (3 - len(self.params)))
        self.c = RglWarp()
        assert L.rgl_warp_init(C.byref(self.c), nx, ny, len(self.params), n_par, par, self.data.ctypes.data_as(fp), int(normalize), int(build_cdf)) == 0

    def _p(self, params):
        return (C.c_double * 3)(*list(params) + [0.0] * (3 - len(params)))

    def eval(self, pos, params=()):
        return _rgl_lib().rgl_warp_eval(C.byref(self.c), (C.c_double * 2)(*pos), self._p(params))

    def sample(self, u, params=()):
        pos = (C.c_double * 2)(); pdf = C.c_double()
        _rgl_lib().rgl_warp_sample(C.byref(self.c), (C.c_double * 2)(*u), self._p(params), pos, C.byref(pdf))
        return (pos[0], pos[1]), pdf.value

    def invert(self, pos, params=()):
        u = (C.c_double * 2)(); pdf = C.c_double()
        _rgl_lib().rgl_warp_invert(C.byref(self.c), (C.c_double * 2)(*pos), self._p(params), u, C.byref(pdf))
        return (u[0], u[1]), pdf.value

    def __del__(self):
        try:
            _rgl_lib().rgl_warp_free(C.byref(self.c))
        except Exception:
            pass


class OracleRgl:
    """The BSDF over the fields of an RGL *.bsdf file (dict of arrays: phi_i, theta_i, ndf, sigma, vndf, luminance, rgb[, jacobian])."""

    def __init__(self, fields):
        """A spectral file holds "spectra" [n_phi, n_theta, n_wavelengths, res, res] and "wavelengths" instead of "rgb"."""
        L = _rgl_lib()
        f32 = lambda k: np.ascontiguousarray(fields[k], np.float32)
        self.spectral = "spectra" in fields
        names = ("phi_i", "theta_i", "ndf", "sigma", "vndf", "luminance") + (("spectra", "wavelengths") if self.spectral else ("rgb",))
        self.f = {k: f32(k) for k in names}
        fp = C.POINTER(C.c_float)
        p = lambda k: self.f[k].ctypes.data_as(fp)
        vn = self.f["vndf"].shape
        jac = int(np.asarray(fields.get("jacobian", 1)).reshape(-1)[0])
        self.c = RglBsdf()
        if self.spectral:
            self.n_wavelengths = int(self.f["wavelengths"].shape[0])
            assert self.f["spectra"].shape == (vn[0], vn[1], self.n_wavelengths, vn[2], vn[3]) and self.f["luminance"].shape == vn
            rc = L.rgl_bsdf_init_spectral(C.byref(self.c), vn[0], vn[1], p("phi_i"), p("theta_i"), self.f["ndf"].shape[1], self.f["ndf"].shape[0], p("ndf"),
                                          self.f["sigma"].shape[1], self.f["sigma"].shape[0], p("sigma"), vn[3], vn[2], p("vndf"), p("luminance"),
                                          self.n_wavelengths, p("wavelengths"), p("spectra"), jac)
        else:
            assert self.f["rgb"].shape == (vn[0], vn[1], 3, vn[2], vn[3]) and self.f["luminance"].shape == vn
            rc = L.rgl_bsdf_init(C.byref(self.c), vn[0], vn[1], p("phi_i"), p("theta_i"), self.f["ndf"].shape[1], self.f["ndf"].shape[0], p("ndf"),
                                 self.f["sigma"].shape[1], self.f["sigma"].shape[0], p("sigma"), vn[3], vn[2], p("vndf"), p("luminance"), p("rgb"), jac)
        assert rc == 0, rc

    def _wl(self, wl, n):
        """(pointer or None, W): per-unit wavelengths [n, W], or None = the file's own nodes"""
        if wl is None:
            return None, None, self.n_wavelengths
        w = np.ascontiguousarray(wl, np.float32)
        assert w.ndim == 2 and w.shape[0] == n
        return w, w.ctypes.data_as(C.POINTER(C.c_float)), int(w.shape[1])

    def eval_pdf_spectral(self, wi, wo, wl=None):
        """values [n, W], pdf [n] of a spectral file at per-unit wavelengths wl [n, W] (None: the file's wavelength nodes)"""
        wi, pwi = _f32(wi); wo, pwo = _f32(wo)
        n = wi.shape[0]
        keep, pwl, W = self._wl(wl, n)
        fp = C.POINTER(C.c_float)
        val = np.empty((n, W), np.float32); pdf = np.empty(n, np.float32)
        _rgl_lib().rgl_eval_pdf_spectral_batch(C.byref(self.c), pwi, pwo, pwl, W, n, val.ctypes.data_as(fp), pdf.ctypes.data_as(fp))
        return val, pdf

    def sample_spectral(self, wi, u, wl=None):
        wi, pwi = _f32(wi); u, pu = _f32(u)
        n = wi.shape[0]
        keep, pwl, W = self._wl(wl, n)
        fp = C.POINTER(C.c_float)
        wo = np.empty((n, 3), np.float32); pdf = np.empty(n, np.float32); w = np.empty((n, W), np.float32)
        _rgl_lib().rgl_sample_spectral_batch(C.byref(self.c), pwi, pu, pwl, W, n, wo.ctypes.data_as(fp), pdf.ctypes.data_as(fp), w.ctypes.data_as(fp))
        return wo, pdf, w

    def eval_pdf(self, wi, wo):
        wi, pwi = _f32(wi); wo, pwo = _f32(wo)
        n = wi.shape[0]
        fp = C.POINTER(C.c_float)
        rgb = np.empty((n, 3), np.float32); pdf = np.empty(n, np.float32)
        _rgl_lib().rgl_eval_pdf_batch(C.byref(self.c), pwi, pwo, n, rgb.ctypes.data_as(fp), pdf.ctypes.data_as(fp))
        return rgb, pdf

    def sample(self, wi, u):
        wi, pwi = _f32(wi); u, pu = _f32(u)
        n = wi.shape[0]
        fp = C.POINTER(C.c_float)
        wo = np.empty((n, 3), np.float32); pdf = np.empty(n, np.float32); w = np.empty((n, 3), np.float32)
        _rgl_lib().rgl_sample_batch(C.byref(self.c), pwi, pu, n, wo.ctypes.data_as(fp), pdf.ctypes.data_as(fp), w.ctypes.data_as(fp))
        return wo, pdf, w

    def half_vector(self, wi, wo):
        """One pair: (wi normalised in the stored part of the azimuth, m = wi + wo unnormalised) as the oracle's f64 arithmetic forms
        them, or None when the pair evaluates to zero."""
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        a = np.ascontiguousarray(wi, np.float32); b = np.ascontiguousarray(wo, np.float32)
        d = np.empty(3, np.float64); m = np.empty(3, np.float64)
        ok = _rgl_lib().rgl_half_vector(C.byref(self.c), a.ctypes.data_as(fp), b.ctypes.data_as(fp), d.ctypes.data_as(dp), m.ctypes.data_as(dp))
        return (d, m) if ok else None

    def eval_pdf_half(self, wi_unit, m):
        """One pair from the normalised incident direction and the unnormalised half vector (f64): (rgb[3], pdf)."""
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        d = np.ascontiguousarray(wi_unit, np.float64); h = np.ascontiguousarray(m, np.float64)
        rgb = np.empty(3, np.float32); pdf = C.c_float()
        _rgl_lib().rgl_eval_pdf_half(C.byref(self.c), d.ctypes.data_as(dp), h.ctypes.data_as(dp), rgb.ctypes.data_as(fp), C.byref(pdf))
        return rgb, np.float32(pdf.value)

    def conditioning_range(self, wi, wo, ulps=8.0):
        """[lo, hi] of (rgb[3], pdf) over the box of half vectors the f64 arithmetic can land on: each of m's transverse components
        carries the rounding of two normalisations and a sum, `ulps` x 1.1e-16 absolute (directions are O(1)).  eval / pdf are not
        multilinear in m (and not monotonic in its azimuth once the box is as large as the transverse length), so the box is
        SAMPLED: corners, edge midpoints, centre and 16 points on its circumscribed circle; None when the pair is zero."""
        hv = self.half_vector(wi, wo)
        if hv is None:
            return None
        d, m = hv
        eta = ulps * 1.1e-16
        offsets = [(dx, dy) for dx in (-eta, 0.0, eta) for dy in (-eta, 0.0, eta)]
        offsets += [(1.4142 * eta * np.cos(a), 1.4142 * eta * np.sin(a)) for a in np.arange(16) * (np.pi / 8) + 0.1]
        vals = []
        for dx, dy in offsets:
            rgb, pdf = self.eval_pdf_half(d, (m[0] + dx, m[1] + dy, m[2]))
            vals.append(np.concatenate([rgb.astype(np.float64), [float(pdf)]]))
        vals = np.array(vals)
        return vals.min(0), vals.max(0)

    def in_conditioning_range(self, what, got, wi, wo):
        """Is `got` — one unit's "eval" (3), "pdf" (1) or "weight" (3: eval / pdf) — where an evaluation of this ill-conditioned pair
        can land?  The sampled range of conditioning_range, widened on either side by a quarter of its width (the samples need not
        hit the extremes) and by 2e-6 relative."""
        r = self.conditioning_range(wi, wo)
        if r is None:
            return False
        lo, hi = r
        if what == "eval":
            lo, hi = lo[:3], hi[:3]
        elif what == "pdf":
            lo, hi = lo[3:4], hi[3:4]
        else:
            lo, hi = lo[:3] / max(hi[3], 1e-300), hi[:3] / max(lo[3], 1e-300)
        g = np.asarray(got, np.float64).reshape(-1)
        slack = 0.25 * (hi - lo) + 2e-6 * np.abs(hi) + 1e-30
        return bool(np.all((g >= lo - slack) & (g <= hi + slack)))

    def __del__(self):
        try:
            _rgl_lib().rgl_bsdf_free(C.byref(self.c))
        except Exception:
            pass
