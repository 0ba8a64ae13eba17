"""The RGL adaptive-parameterisation BSDF (the "*.bsdf tensor format of upstream M3 `measured`") — the
oracle's restatement (oracle/rgl_oracle.c) pinned by self-consistency on the CPU.  PARITY UNPINNED: no measured file, no upstream
plugin and no reference vector exist in the container; what is checked is that the restated model is a consistent
distribution machinery — the piecewise-bilinear warp's sample / invert / eval agree with each other and with closed forms, the
BSDF's pdf is a density, sample() reports it, weight == eval / pdf."""
import numpy as np
import pytest

from mitsuba_customization_amd import synth
from oracle import binding as ob


def test_constant_warp_is_the_identity():
    w = ob.OracleWarp(np.ones((5, 7), np.float32))
    for u in ((0.3, 0.8), (0.0, 0.0), (0.999, 0.001), (0.5, 0.5)):
        pos, pdf = w.sample(u)
        assert np.allclose(pos, u, atol=1e-7) and abs(pdf - 1) < 1e-6
        assert np.allclose(w.invert(u)[0], u, atol=1e-7) and abs(w.eval(u) - 1) < 1e-6


@pytest.mark.parametrize("params", [(), (0.3, 0.6)])
def test_sample_and_invert_are_inverse_and_agree_with_eval(params):
    rng = np.random.default_rng(3)
    grids = [np.array([0.0, 0.5, 1.0], np.float32), np.array([0.0, 0.2, 0.7, 1.0], np.float32)][:len(params)]
    shape = tuple(len(g) for g in grids) + (9, 13)
    w = ob.OracleWarp((rng.random(shape) + 0.05).astype(np.float32), grids)
    for u in rng.random((400, 2)):
        pos, pdf = w.sample(u, params)
        back, pdf2 = w.invert(pos, params)
        assert np.allclose(back, u, atol=2e-9), (u, pos, back)
        assert abs(pdf - pdf2) <= 1e-12 * pdf and abs(pdf - w.eval(pos, params)) <= 1e-12 * pdf
        assert 0 <= pos[0] <= 1 and 0 <= pos[1] <= 1
    # the density integrates to one at interpolated parameters: a bilinear patch integrates to the mean of its corners
    ny, nx = 9, 13
    tot = sum(0.25 * (w.eval((x / (nx - 1), y / (ny - 1)), params) + w.eval(((x + 1) / (nx - 1), y / (ny - 1)), params) +
                      w.eval((x / (nx - 1), (y + 1) / (ny - 1)), params) + w.eval(((x + 1) / (nx - 1), (y + 1) / (ny - 1)), params))
              for x in range(nx - 1) for y in range(ny - 1)) / ((nx - 1) * (ny - 1))
    assert abs(tot - 1) < 2e-6


def test_unnormalised_warp_is_bilinear_interpolation_of_the_data():
    rng = np.random.default_rng(9)
    d = rng.random((6, 5)).astype(np.float32)
    w = ob.OracleWarp(d, normalize=False, build_cdf=False)
    for x, y in rng.random((100, 2)):
        px, py = x * 4, y * 5
        i, j = min(int(px), 3), min(int(py), 4)
        fx, fy = px - i, py - j
        want = (1 - fy) * ((1 - fx) * d[j, i] + fx * d[j, i + 1]) + fy * ((1 - fx) * d[j + 1, i] + fx * d[j + 1, i + 1])
        assert abs(w.eval((x, y)) - want) < 1e-12
    # a parameter exactly on a grid node selects that slice (the rgb channel axis works that way)
    s = rng.random((3, 4, 4)).astype(np.float32)
    w3 = ob.OracleWarp(s, [np.array([0, 1, 2], np.float32)], normalize=False, build_cdf=False)
    for c in range(3):
        assert abs(w3.eval((0.0, 0.0), (float(c),)) - s[c, 0, 0]) < 1e-12


@pytest.mark.parametrize("n_phi,reduction", [(1, 1), (5, 1), (4, 2), (3, 4)])
def test_bsdf_pdf_is_a_density_and_sample_reports_it(n_phi, reduction):
    B = ob.OracleRgl(synth.make_rgl_fields(4, n_phi=n_phi, reduction=reduction))
    rng = np.random.default_rng(21)
    for mu, az in ((0.95, 0.3), (0.6, -2.0), (0.25, 1.1)):
        wi1 = np.array([np.sqrt(1 - mu * mu) * np.cos(az), np.sqrt(1 - mu * mu) * np.sin(az), mu], np.float32)
        nz, nphi = 300, 360
        z = (np.arange(nz) + 0.5) / nz
        ph = (np.arange(nphi) + 0.5) / nphi * 2 * np.pi - np.pi
        Z, P = np.meshgrid(z, ph, indexing="ij")
        r = np.sqrt(1 - Z * Z)
        wo = np.stack([r * np.cos(P), r * np.sin(P), Z], -1).reshape(-1, 3).astype(np.float32)
        rgb, pdf = B.eval_pdf(np.tile(wi1, (wo.shape[0], 1)), wo)
        assert (rgb >= 0).all() and np.isfinite(rgb).all() and (pdf >= 0).all()
        u = rng.random((200000, 2)).astype(np.float32)
        s_wo, s_pdf, s_w = B.sample(np.tile(wi1, (u.shape[0], 1)), u)
        live = s_pdf > 0
        # The pdf over directions has an integrable singularity at the mirror direction (the map u_m -> wo has Jacobian
        # 2 pi^2 u sin(theta_m) 4 wi.m -> 0), which a quadrature over directions under-integrates; in the domain of the warp the
        # same integral is smooth: the mass of u_m whose reflected direction is above the horizon == the accepted fraction.
        import ctypes as C
        L = ob._rgl_lib()
        wi_d = wi1.astype(np.float64) / np.linalg.norm(wi1.astype(np.float64))
        if reduction >= 2:                                          # into the stored part of the azimuth (the mass is the same there)
            sy = 1.0 if np.signbit(wi_d[1]) else -1.0
            sx = (1.0 if np.signbit(wi_d[0]) else -1.0) if reduction == 4 else sy
            wi_d = wi_d * np.array([sx, sy, 1.0])
        theta_i = 2 * np.arcsin(min(1.0, 0.5 * np.linalg.norm(wi_d - np.array([0, 0, 1.0]))))
        phi_i = np.arctan2(wi_d[1], wi_d[0])
        par = (C.c_double * 3)(phi_i, theta_i, 0.0)
        G = 160
        mass = 0.0
        for a in range(G):
            for b in range(G):
                um = ((a + 0.5) / G, (b + 0.5) / G)
                theta_m = um[0] ** 2 * np.pi / 2
                phi_m = (2 * um[1] - 1) * np.pi + (phi_i if n_phi <= 2 else 0.0)
                m = np.array([np.cos(phi_m) * np.sin(theta_m), np.sin(phi_m) * np.sin(theta_m), np.cos(theta_m)])
                c = float(wi_d @ m)
                if c <= 0 or (2 * c * m - wi_d)[2] <= 0:
                    continue
                smp = (C.c_double * 2)(); vp = C.c_double()
                L.rgl_warp_invert(C.byref(B.c.vndf), (C.c_double * 2)(*um), par, smp, C.byref(vp))
                mass += vp.value * L.rgl_warp_eval(C.byref(B.c.luminance), smp, par)
        mass /= G * G
        assert abs(mass - live.mean()) < 0.01, (mu, mass, live.mean())
        # sample() reports pdf(wi, wo') and weight == eval / pdf, exactly
        e_rgb, e_pdf = B.eval_pdf(np.tile(wi1, (int(live.sum()), 1)), s_wo[live])
        assert np.array_equal(e_pdf, s_pdf[live]) and np.array_equal(s_w[live], (e_rgb / e_pdf[:, None]).astype(np.float32))
        # the estimator's mean equals the quadrature of eval (f cos) over the hemisphere
        albedo_q = rgb.astype(np.float64).mean(axis=0) * 2 * np.pi
        albedo_s = s_w.astype(np.float64).mean(axis=0)
        assert np.allclose(albedo_q, albedo_s, rtol=0.03), (albedo_q, albedo_s)


def test_guards():
    B = ob.OracleRgl(synth.make_rgl_fields(1))
    wi = np.array([[0, 0, -1], [0.3, 0.1, 0.9]], np.float32)
    wo = np.array([[0.1, 0.2, 0.9], [0.1, 0.2, -0.9]], np.float32)
    rgb, pdf = B.eval_pdf(wi, wo)
    assert not rgb.any() and not pdf.any()
    s_wo, s_pdf, s_w = B.sample(wi[:1], np.array([[0.3, 0.4]], np.float32))
    assert not s_wo.any() and not s_pdf.any() and not s_w.any()


@pytest.mark.skipif(__import__("shutil").which("hipcc") is None, reason="hipcc missing")
@pytest.mark.parametrize("case", [dict(seed=1, n_phi=1, n_theta=6, res=12, res_ndf=16, res_sigma=8),
                                  dict(seed=2, n_phi=5, n_theta=4, res=9, res_ndf=8, res_sigma=6),
                                  dict(seed=3, n_phi=1, n_theta=1, res=2, res_ndf=2, res_sigma=2),
                                  dict(seed=4, n_phi=4, n_theta=3, res=7, res_ndf=8, res_sigma=6, reduction=2),
                                  dict(seed=5, n_phi=3, n_theta=3, res=6, res_ndf=6, res_sigma=4, reduction=4),
                                  dict(seed=6, n_phi=5, n_theta=1, res=8, res_ndf=8, res_sigma=6)],      # one elevation node: a bracket of two phi slices
                         ids=lambda c: f"phi{c['n_phi']}_theta{c['n_theta']}_res{c['res']}_red{c.get('reduction', 1)}")
def test_product_per_unit_functions_on_the_host_match_the_oracle(case, tmp_path_factory):
    """The product's RGL code — image builder (cell bricks, running integrals) and per-unit eval / pdf / sample, the SAME
    functions the kernel runs (__host__ __device__) — compiled for the host (tests/rgl_host_harness.hip) against the
    independent restatement in oracle/rgl_oracle.c.  The product uses reciprocal / rsqrt-seeded Newton steps, an atan
    polynomial and Taylor sin / cos where the oracle calls libm: 1e-6 relative."""
    import os
    import subprocess
    from oracle.binding import OracleRgl, generate_pairs
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    build = tmp_path_factory.getbasetemp() / "rgl_host_harness"
    if not build.exists():
        subprocess.check_call(["hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-mavx2", "-mfma", "-w", "-o", str(build),
                               os.path.join(root, "tests", "rgl_host_harness.hip")])
    tmp = tmp_path_factory.mktemp("rgl_host")
    f = synth.make_rgl_fields(**case)
    with open(tmp / "f.bin", "wb") as o:
        np.array([case["n_phi"], case["n_theta"], case["res"], case["res_ndf"], case["res_sigma"], 1], np.int32).tofile(o)
        for k in ("phi_i", "theta_i", "ndf", "sigma", "vndf", "luminance", "rgb"):
            f[k].astype(np.float32).tofile(o)
    n = 30000
    wi, wo, u = generate_pairs(7 + case["seed"], 0, n)
    wi[:50] = wi[50:100]; wo[:50] = wi[:50] * np.array([-1, -1, 1], np.float32)          # exact mirror pairs: m == n
    wo[100:150] = wi[100:150] * np.array([-1, -1, 1], np.float32) * np.float32(1 + 3e-7)   # ... and pairs a few ulps off
    with open(tmp / "p.bin", "wb") as o:
        np.array([n], np.uint64).tofile(o); wi.tofile(o); wo.tofile(o); u.tofile(o)
    r = subprocess.run([str(build), str(tmp / "f.bin"), str(tmp / "p.bin"), str(tmp / "o.bin")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(tmp / "o.bin", np.float32).reshape(n, 11)
    orc = OracleRgl(f)

    def close(a, b, what):
        b = np.asarray(b, np.float64)
        err = np.abs(a.astype(np.float64) - b) / (np.abs(b) + 0.1 * max(float(np.abs(b).max()), 1e-30))
        assert float(err.max()) < 1e-6, (what, float(err.max()), int(err.argmax()))

    rgb, pdf = orc.eval_pdf(wi, wo)
    close(out[:, 0:3], rgb, "eval"); close(out[:, 3], pdf, "pdf")
    o_wo, o_pdf, _ = orc.sample(wi, u)
    live = out[:, 7] > 0
    assert live.mean() > 0.5 and np.count_nonzero(live != (o_pdf > 0)) <= 2
    both = live & (o_pdf > 0)
    assert float(np.abs(out[both, 4:7] - o_wo[both]).max()) < 5e-7
    c_rgb, c_pdf = orc.eval_pdf(wi[live], out[live, 4:7])
    close(out[live, 7], c_pdf, "sample pdf"); close(out[live, 8:11], c_rgb / c_pdf[:, None], "sample weight")


@pytest.mark.parametrize("reduction", [2, 4])
def test_symmetry_reduced_files_are_equivariant(reduction):
    """A file that stores phi_i in [-pi, 0] (reduction 2: point symmetry) or [-pi, -pi/2] (reduction 4: two mirror planes) answers for
    the whole azimuth: eval / pdf are invariant under the sample's symmetry operations and sample() is equivariant (the direction
    drawn for S wi is S applied to the direction drawn for wi) — which is what pins the map back out of the stored part."""
    B = ob.OracleRgl(synth.make_rgl_fields(6, n_phi=4, n_theta=4, res=8, reduction=reduction))
    assert B.c.reduction == reduction
    wi, wo, u = ob.generate_pairs(77, 0, 4000)
    ops = [np.array([-1, -1, 1], np.float32)] + ([np.array([-1, 1, 1], np.float32), np.array([1, -1, 1], np.float32)] if reduction == 4 else [])
    rgb, pdf = B.eval_pdf(wi, wo)
    s_wo, s_pdf, s_w = B.sample(wi, u)
    assert (pdf > 0).mean() > 0.9 and (s_pdf > 0).mean() > 0.5
    for S in ops:
        r2, p2 = B.eval_pdf(wi * S, wo * S)
        assert np.array_equal(r2, rgb) and np.array_equal(p2, pdf)
        w2, q2, v2 = B.sample(wi * S, u)
        assert np.array_equal(w2, s_wo * S) and np.array_equal(q2, s_pdf) and np.array_equal(v2, s_w)
    # and the stored part is really what is read: wi inside it is not touched
    inside = (wi[:, 1] < 0) & ((wi[:, 0] < 0) | (reduction == 2))
    B1 = ob.OracleRgl(dict(synth.make_rgl_fields(6, n_phi=4, n_theta=4, res=8, reduction=reduction)))
    assert inside.any() and np.array_equal(B1.eval_pdf(wi[inside], wo[inside])[0], rgb[inside])
    # a span that is no integer fraction of the circle is refused
    bad = synth.make_rgl_fields(6, n_phi=4, n_theta=4, res=8, reduction=2)
    bad["phi_i"] = np.linspace(-np.pi, -1.0, 4).astype(np.float32)
    with pytest.raises(AssertionError):
        ob.OracleRgl(bad)


# ------------------------------------------------------------------ spectral files ("spectra" + "wavelengths" instead of "rgb")
def test_spectral_files_interpolate_the_wavelength_as_a_third_parameter():
    """What pins the spectral restatement (no spectral file or upstream source exists offline: PARITY UNPINNED):
    at the file's own nodes the values are the per-node tables' (a spectral file whose three "wavelengths" are 0, 1, 2 IS the RGB file,
    bit for bit); between two nodes the value is their linear blend; outside the grid it is the end node's; pdf and sampled direction do
    not depend on the wavelength and equal the RGB file's; weight == value / pdf."""
    from mitsuba_customization_amd import synth
    from oracle import binding as ob
    rgbf = synth.make_rgl_fields(seed=31, n_phi=1, n_theta=4, res=7, res_ndf=6, res_sigma=5)
    spec = {k: v for k, v in rgbf.items() if k != "rgb"}
    spec["spectra"] = rgbf["rgb"].copy(); spec["wavelengths"] = np.array([0.0, 1.0, 2.0], np.float32)
    A, B = ob.OracleRgl(rgbf), ob.OracleRgl(spec)
    wi, wo, u = ob.generate_pairs(0x5EED, 31, 4000)
    rgb, pdf = A.eval_pdf(wi, wo)
    val, pdf_s = B.eval_pdf_spectral(wi, wo)                     # at the nodes
    assert np.array_equal(val, rgb) and np.array_equal(pdf_s, pdf)
    wo2, pdf2, w = A.sample(wi, u)
    s_wo2, s_pdf2, s_w = B.sample_spectral(wi, u)
    assert np.array_equal(s_wo2, wo2) and np.array_equal(s_pdf2, pdf2) and np.array_equal(s_w, w)
    # per-unit wavelengths: midpoints blend, outside clamps
    n = wi.shape[0]
    wl = np.tile(np.array([0.5, 1.25, -3.0, 7.0, 2.0], np.float32), (n, 1))
    v, p = B.eval_pdf_spectral(wi, wo, wl)
    r64 = rgb.astype(np.float64)
    assert np.array_equal(p, pdf)
    assert np.allclose(v[:, 0], 0.5 * (r64[:, 0] + r64[:, 1]), rtol=3e-7, atol=1e-30)
    assert np.allclose(v[:, 1], 0.75 * r64[:, 1] + 0.25 * r64[:, 2], rtol=3e-7, atol=1e-30)
    assert np.array_equal(v[:, 2], rgb[:, 0]) and np.array_equal(v[:, 3], rgb[:, 2]) and np.array_equal(v[:, 4], rgb[:, 2])
    s2_wo, s2_pdf, s2_w = B.sample_spectral(wi, u, wl)
    assert np.array_equal(s2_wo, wo2) and np.array_equal(s2_pdf, pdf2)
    live = s2_pdf > 0
    c_val, c_pdf = B.eval_pdf_spectral(wi[live], s2_wo[live], wl[live])
    assert np.array_equal(c_pdf, s2_pdf[live]) and np.array_equal(s2_w[live], c_val / c_pdf[:, None])


@pytest.mark.skipif(__import__("shutil").which("hipcc") is None, reason="hipcc missing")
def test_product_spectral_functions_on_the_host_match_the_oracle(tmp_path_factory):
    """The product's spectral per-unit functions and image builder compiled for the HOST (tests/rgl_host_harness.hip --spectral)
    against the oracle: values / weights to 1e-6, pdf to 1e-6, directions to an ulp."""
    import os, subprocess
    from mitsuba_customization_amd import synth
    from oracle import binding as ob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    build = tmp_path_factory.getbasetemp() / "rgl_host_harness"
    if not build.exists():
        subprocess.check_call(["hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-mavx2", "-mfma", "-w", "-o", str(build),
                               os.path.join(root, "tests", "rgl_host_harness.hip")])
    tmp = tmp_path_factory.mktemp("spectral")
    for case in (dict(seed=41, n_phi=1, n_theta=5, res=9, res_ndf=8, res_sigma=6, n_wavelengths=7),
                 dict(seed=42, n_phi=4, n_theta=3, res=6, res_ndf=6, res_sigma=4, n_wavelengths=1),
                 dict(seed=43, n_phi=5, n_theta=2, res=5, res_ndf=6, res_sigma=4, n_wavelengths=12)):
        f = synth.make_rgl_fields(**case)
        B = ob.OracleRgl(f)
        n, W = 3000, 4
        wi, wo, u = ob.generate_pairs(0x5EED, case["seed"], n)
        rng = np.random.default_rng(case["seed"])
        wl = rng.uniform(300.0, 1060.0, (n, W)).astype(np.float32)            # some outside the grid
        with open(tmp / "fields.bin", "wb") as fh:
            np.array([case["n_phi"], case["n_theta"], case["res"], case["res_ndf"], case["res_sigma"], 1, case["n_wavelengths"]], np.int32).tofile(fh)
            for k in ("phi_i", "theta_i", "ndf", "sigma", "vndf", "luminance", "spectra", "wavelengths"):
                np.ascontiguousarray(f[k], np.float32).tofile(fh)
        with open(tmp / "pairs.bin", "wb") as fh:
            np.array([n], np.uint64).tofile(fh); np.array([W], np.int32).tofile(fh)
            wi.tofile(fh); wo.tofile(fh); u.tofile(fh); wl.tofile(fh)
        subprocess.check_call([str(build), "--spectral", str(tmp / "fields.bin"), str(tmp / "pairs.bin"), str(tmp / "out.bin")])
        out = np.fromfile(tmp / "out.bin", np.float32).reshape(n, 2 * W + 5)
        val, pdf, wo2, pdf2, w = out[:, :W], out[:, W], out[:, W + 1:W + 4], out[:, W + 4], out[:, W + 5:]
        o_val, o_pdf = B.eval_pdf_spectral(wi, wo, wl)
        close = lambda a, b: bool((np.abs(a.astype(np.float64) - b) <= 1e-6 * np.abs(b) + 1e-30).all())
        assert close(val, o_val) and close(pdf, o_pdf), case
        o_wo2, o_pdf2, _ = B.sample_spectral(wi, u, wl)
        live = (pdf2 > 0) & (o_pdf2 > 0)
        assert np.count_nonzero((pdf2 > 0) != (o_pdf2 > 0)) <= 1 and float(np.abs(wo2[live] - o_wo2[live]).max()) < 5e-7
        c_val, c_pdf = B.eval_pdf_spectral(wi[live], wo2[live], wl[live])
        assert close(pdf2[live], c_pdf) and close(w[live], c_val / c_pdf[:, None]), case
