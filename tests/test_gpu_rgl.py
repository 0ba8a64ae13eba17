"""The adaptive-parameterisation measured BSDF (RGL *.bsdf fields; upstream Mitsuba 3 `measured`) on the GPU against the
CPU restatement oracle/rgl_oracle.c.  PARITY UNPINNED: no RGL file and no upstream source exist offline; the oracle is pinned
by the self-consistency KATs of tests/test_rgl_cpu.py only, and the tables are synthetic (synth.make_rgl_fields).
Tolerance: north_star's — |gpu - oracle| <= 1e-6 |oracle| for EVERY value (both sides compute in f64 on the same Float tables
and differ by FMA use and libm-vs-polynomial rounding before one rounding to Float).  The one family that cannot meet it is
ill-conditioned on BOTH sides: for a near-mirror pair the half vector's transverse part is the difference of two
normalisations (1e-16 of rounding on a length that can be 1e-9), its azimuth then carries a handful of significant bits
whichever arithmetic forms it.  A value outside 1e-6 must lie inside the range the ORACLE spans over that rounding box
(OracleRgl.in_conditioning_range: 8 f64 ulps on the half vector's transverse components, 25 sample points, widened by a
quarter of its width — the analogue of test_gpu_parity.py::_conditioning_range for the tables)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [dict(seed=1, n_phi=1, n_theta=6, res=12, res_ndf=16, res_sigma=8),          # isotropic, the common case
         dict(seed=2, n_phi=5, n_theta=4, res=9, res_ndf=8, res_sigma=6),            # anisotropic
         dict(seed=3, n_phi=1, n_theta=1, res=2, res_ndf=2, res_sigma=2),            # the smallest legal file
         dict(seed=4, n_phi=4, n_theta=3, res=7, res_ndf=8, res_sigma=6, reduction=2),   # phi_i in [-pi, 0]: point symmetry
         dict(seed=5, n_phi=3, n_theta=3, res=6, res_ndf=6, res_sigma=4, reduction=4),   # phi_i in [-pi, -pi/2]: two mirror planes
         dict(seed=6, n_phi=5, n_theta=1, res=8, res_ndf=8, res_sigma=6)]                # one elevation node: brackets of two phi slices (the kernel that tests the shape at run time)


def _close(a, b, what, orc=None, wi=None, wo=None, max_ill=0):
    """Every value within 1e-6 relative of the oracle's.  With (orc, wi, wo): a unit that is not must be ill-conditioned — all of
    its values inside the oracle's own rounding range — and there may be at most max_ill such units.  `what`: "eval" (n x 3),
    "pdf" (n), "weight" (n x 3: eval / pdf)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    ok = np.abs(a - b) <= 1e-6 * np.abs(b) + 1e-30
    if ok.all():
        return 0
    rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-30)
    assert orc is not None, (what, float(rel.max()), int(rel.argmax()))
    bad = np.nonzero(~ok.reshape(a.shape[0], -1).all(axis=1))[0]
    assert bad.size <= max_ill, (what, bad.size, float(rel.max()))
    kind = "pdf" if "pdf" in what else what
    for i in bad:
        assert orc.in_conditioning_range(kind, a[i], wi[i], wo[i]), (what, int(i), a[i].tolist(), b[i].tolist(), orc.conditioning_range(wi[i], wo[i]))
    return int(bad.size)


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"phi{c['n_phi']}_theta{c['n_theta']}_res{c['res']}_red{c.get('reduction', 1)}")
def test_eval_pdf_sample_match_the_oracle(case):
    from mitsuba_customization_amd import host, synth
    from oracle.binding import OracleRgl
    fields = synth.make_rgl_fields(**case)
    orc = OracleRgl(fields)
    n = 1 << 15
    with host.MerlHip(0) as g:
        mid = g.upload_rgl(fields)
        kind, dims = g.material_info(mid)
        assert kind == host.KIND_RGL and dims == (case["n_phi"], case["n_theta"], case["res"])
        wi_t, wo_t, u_t = g.generate_pairs(0x861 + case["seed"], 0, n)
        wi, wo, u = wi_t.cpu().numpy(), wo_t.cpu().numpy(), u_t.cpu().numpy()
        rgb = g.eval(wi_t, wo_t, material=mid).cpu().numpy()
        pdf = g.pdf(wi_t, wo_t, material=mid).cpu().numpy()
        o_rgb, o_pdf = orc.eval_pdf(wi, wo)
        assert float(o_rgb.max()) > 0 and float(o_pdf.max()) > 0
        _close(rgb, o_rgb, "eval", orc, wi, wo, max_ill=2); _close(pdf, o_pdf, "pdf", orc, wi, wo, max_ill=2)
        wo2, pdf2, w = (t.cpu().numpy() for t in g.sample(wi_t, u_t, material=mid))
        o_wo2, o_pdf2, o_w = orc.sample(wi, u)
        live = o_pdf2 > 0
        assert live.mean() > 0.5
        assert np.array_equal(pdf2 > 0, live) or (np.count_nonzero((pdf2 > 0) != live) <= 2)     # a sample on the horizon may round either way
        both = live & (pdf2 > 0)
        assert float(np.abs(wo2[both] - o_wo2[both]).max()) < 5e-7                     # the same Float direction to an ulp
        # eval / pdf are steep functions of direction near the specular peak: compare what each side reports AT ITS OWN direction
        # with the oracle evaluated there
        c_rgb, c_pdf = orc.eval_pdf(wi[both], wo2[both])
        _close(pdf2[both], c_pdf, "sample pdf", orc, wi[both], wo2[both], max_ill=2)
        _close(w[both], c_rgb / c_pdf[:, None], "weight", orc, wi[both], wo2[both], max_ill=2)
        # the fused entry points agree bit for bit with the separate ones
        f_rgb, f_pdf = g.eval_pdf(wi_t, wo_t, material=mid)
        assert np.array_equal(f_rgb.cpu().numpy().view(np.int32), rgb.view(np.int32)) and np.array_equal(f_pdf.cpu().numpy().view(np.int32), pdf.view(np.int32))
        es = [t.cpu().numpy() for t in g.eval_sample(wi_t, wo_t, u_t, material=mid)]
        for got, want in zip(es, (rgb, pdf, wo2, pdf2, w)):
            assert np.array_equal(got.view(np.int32), want.view(np.int32))
        # the search tables read from memory (bracket form) instead of the copy in LDS: the same bits from every entry point
        assert g.get_option(host.OPT_RGL_SEARCH) == 0
        g.set_option(host.OPT_RGL_SEARCH, 1)
        for got, want in zip([t.cpu().numpy() for t in g.eval_sample(wi_t, wo_t, u_t, material=mid)], (rgb, pdf, wo2, pdf2, w)):
            assert np.array_equal(got.view(np.int32), want.view(np.int32))
        for got, want in zip([t.cpu().numpy() for t in g.sample(wi_t, u_t, material=mid)], (wo2, pdf2, w)):
            assert np.array_equal(got.view(np.int32), want.view(np.int32))
        assert np.array_equal(g.eval(wi_t, wo_t, material=mid).cpu().numpy().view(np.int32), rgb.view(np.int32))
        assert np.array_equal(g.pdf(wi_t, wo_t, material=mid).cpu().numpy().view(np.int32), pdf.view(np.int32))
        g.set_option(host.OPT_RGL_SEARCH, 0)
        # pdf(wi, sample.wo) == sample.pdf, weight == eval / pdf — on the device's own outputs
        import torch
        wo2_t = torch.from_numpy(wo2).to(wi_t.device)
        back = g.pdf(wi_t, wo2_t, material=mid).cpu().numpy()
        assert np.array_equal(back[pdf2 > 0].view(np.int32), pdf2[pdf2 > 0].view(np.int32))
        if case.get("reduction", 1) > 1:               # the sample's symmetry: invariant eval / pdf, equivariant sample, bit for bit
            ops = [(-1, -1, 1)] + ([(-1, 1, 1), (1, -1, 1)] if case["reduction"] == 4 else [])
            for op in ops:
                S = torch.tensor(op, device=wi_t.device, dtype=torch.float32)
                r2, p2 = g.eval_pdf(wi_t * S, wo_t * S, material=mid)
                assert torch.equal(r2.cpu().view(torch.int32), torch.from_numpy(rgb).view(torch.int32)) and torch.equal(p2.cpu(), torch.from_numpy(pdf))
                w2, q2, v2 = g.sample(wi_t * S, u_t, material=mid)
                assert torch.equal(w2.cpu(), torch.from_numpy(wo2) * S.cpu()) and torch.equal(q2.cpu(), torch.from_numpy(pdf2))
        # lower hemisphere: zeros
        down = wi.copy(); down[:, 2] = -np.abs(down[:, 2])
        z = g.eval(torch.from_numpy(down).to(wi_t.device), wo_t, material=mid)
        assert float(z.abs().max()) == 0.0


def test_host_arrays_queues_and_mixed_batches():
    import torch
    from mitsuba_customization_amd import host, synth
    fields = synth.make_rgl_fields(seed=4, n_phi=1, n_theta=5, res=10)
    n = 5000
    with host.MerlHip(0) as g:
        ggx = g.ggx(0.2, (1.5, 1.5, 1.5), (3.0, 3.0, 3.0))
        mid = g.upload_rgl(fields)
        wi_t, wo_t, u_t = g.generate_pairs(77, 0, n)
        dev = [t.cpu().numpy() for t in g.eval_sample(wi_t, wo_t, u_t, material=mid)]
        # host arrays (pipelined staging) give the same bits
        hst = g.eval_sample(wi_t.cpu().numpy(), wo_t.cpu().numpy(), u_t.cpu().numpy(), material=mid)
        for a, b in zip(hst, dev):
            assert np.array_equal(np.asarray(a).view(np.int32), b.view(np.int32))
        # a queue of every other unit: those slots are written, the rest keep their sentinel
        q = torch.arange(0, n, 2, device=wi_t.device, dtype=torch.int32)
        cnt = torch.tensor([q.numel()], device=wi_t.device, dtype=torch.int32)
        outq = g.eval_sample_queue(wi_t, wo_t, u_t, q, cnt, material=mid)
        assert torch.equal(outq[0][0::2].cpu(), torch.from_numpy(dev[0][0::2]))
        # mixed batches: tables, an analytic material and two RGL materials side by side, every unit the bits of its material's own call
        from mitsuba_customization_amd import synth as sy
        tab = g.upload_table(sy.make_table("noise", 3, (8, 8, 16)), (1.0, 1.0, 1.0))
        mid2 = g.upload_rgl(sy.make_rgl_fields(seed=8, n_phi=5, n_theta=3, res=6))
        ids = torch.tensor([ggx, mid, tab, mid2, 99, -1], device=wi_t.device, dtype=torch.int32)
        mat = ids[torch.arange(n, device=wi_t.device) % ids.numel()]
        alone = {int(k): g.eval_sample(wi_t, wo_t, u_t, material=int(k)) for k in (ggx, mid, tab, mid2)}
        for variant in (0, 1, 2, 3, 4):
            g.set_option(host.OPT_KERNEL, variant)
            mixed = g.eval_sample(wi_t, wo_t, u_t, mat=mat)
            for slot, k in enumerate(ids.tolist()):
                for got, want in zip(mixed, alone.get(k, [None] * 5)):
                    if want is None:
                        assert float(got[slot::6].abs().max()) == 0.0                      # unknown ids: zeros
                    else:
                        assert torch.equal(got[slot::6].view(torch.int32), want[slot::6].view(torch.int32)), (variant, k)
        g.set_option(host.OPT_KERNEL, 3)
        for call, single in ((lambda **kw: (g.eval(wi_t, wo_t, **kw),), 0), (lambda **kw: (g.pdf(wi_t, wo_t, **kw),), 1),
                             (lambda **kw: g.sample(wi_t, u_t, **kw), 2), (lambda **kw: g.eval_pdf(wi_t, wo_t, **kw), 4)):
            mixed = call(mat=mat)
            for slot, k in ((1, mid), (3, mid2), (0, ggx)):
                for got, want in zip(mixed, call(material=int(k))):
                    assert torch.equal(got[slot::6].view(torch.int32), want[slot::6].view(torch.int32)), (single, k)
        # a queue over a mixed batch, and host arrays
        qm = torch.arange(1, n, 3, device=wi_t.device, dtype=torch.int32)
        cm = torch.tensor([qm.numel()], device=wi_t.device, dtype=torch.int32)
        full = g.eval_sample(wi_t, wo_t, u_t, mat=mat)
        outm = g.eval_sample_queue(wi_t, wo_t, u_t, qm, cm, mat=mat)
        for got, want in zip(outm, full):
            assert torch.equal(got[1::3].view(torch.int32), want[1::3].view(torch.int32)) and float(got[0::3].abs().max()) == 0.0
        hm = g.eval_sample(wi_t.cpu().numpy(), wo_t.cpu().numpy(), u_t.cpu().numpy(), mat=mat.cpu().numpy())
        for got, want in zip(hm, full):
            assert np.array_equal(np.asarray(got).view(np.int32), want.cpu().numpy().view(np.int32))
        g.release_material(tab); g.release_material(mid2)
        # the device's one-unit call service does not take it ...
        with pytest.raises(host.MerlHipError) as e:
            g.scalar_eval_sample(wi_t[0].cpu().numpy(), wo_t[0].cpu().numpy(), u_t[0].cpu().numpy(), material=mid)
        assert e.value.status == host.ERR_MATERIAL
        # ... one-unit calls run on the CPU over a host image: the kernels' own per-unit functions compiled for the host
        with g.host_table(mid) as h:
            assert h.info()["bytes"] == g.memory_info()["table_bytes"]
            wi_h, wo_h, u_h = wi_t.cpu().numpy(), wo_t.cpu().numpy(), u_t.cpu().numpy()
            one = np.stack([h.eval_sample(wi_h[i], wo_h[i], u_h[i]) for i in range(512)])
        want = np.concatenate([dev[0][:512], dev[1][:512, None], dev[2][:512], dev[3][:512, None], dev[4][:512]], axis=1)
        assert np.allclose(one, want, rtol=1e-6, atol=1e-7 * float(np.abs(want).max()))
        assert np.mean(one.view(np.int32) == want.view(np.int32)) > 0.95
        # release gives the memory back and the slot is reusable
        used = g.memory_info()["table_bytes"]
        assert used > 0
        g.release_material(mid)
        assert g.memory_info()["table_bytes"] == 0
        assert g.upload_rgl(fields) == mid


def test_load_from_a_tensor_file_and_refusals(tmp_path):
    from mitsuba_customization_amd import host, synth
    fields = synth.make_rgl_fields(seed=5, n_phi=1, n_theta=4, res=8)
    path = str(tmp_path / "synthetic_rgb.bsdf")
    synth.write_tensor_file(path, fields)
    with host.MerlHip(0) as g:
        a = g.load_rgl(path)
        b = g.upload_rgl(fields)
        wi, wo, u = g.generate_pairs(5, 0, 4096)
        for x, y in zip(g.eval_sample(wi, wo, u, material=a), g.eval_sample(wi, wo, u, material=b)):
            assert np.array_equal(x.cpu().numpy().view(np.int32), y.cpu().numpy().view(np.int32))
        count = g.material_count()

        def refused(mutate, needle):
            f = dict(fields); mutate(f)
            p = str(tmp_path / "bad.bsdf")
            synth.write_tensor_file(p, f)
            with pytest.raises(host.MerlHipError) as e:
                g.load_rgl(p)
            assert needle in str(e.value), str(e.value)
            assert g.material_count() == count

        refused(lambda f: f.pop("vndf"), "vndf")
        refused(lambda f: (f.pop("rgb"), f.__setitem__("spectra", np.zeros((1, 4, 5, 8, 8), np.float32))), "wavelengths")   # a spectral file needs its grid (test_gpu_rgl_spectral.py)
        refused(lambda f: f.__setitem__("rgb", f["rgb"][:, :, :2]), "rgb")
        refused(lambda f: f.__setitem__("theta_i", f["theta_i"][::-1].copy()), "ascending")
        refused(lambda f: f.__setitem__("vndf", -f["vndf"]), "non-negative")
        refused(lambda f: f.__setitem__("ndf", f["ndf"].astype(np.float64)), "float32")
        bad = f = dict(fields); nan = fields["luminance"].copy(); nan[0, 0, 0, 0] = np.nan
        refused(lambda f: f.__setitem__("luminance", nan), "non-finite")
        # an anisotropic file stores the whole azimuth, a half or a quarter of it: anything else is refused
        aniso = synth.make_rgl_fields(seed=6, n_phi=4, n_theta=3, res=6)
        aniso["phi_i"] = np.linspace(-np.pi, -1.0, 4).astype(np.float32)
        with pytest.raises(host.MerlHipError) as e:
            g.upload_rgl(aniso)
        assert "azimuth" in str(e.value)


@pytest.mark.parametrize("name", ["rgl_isotropic", "rgl_anisotropic"])
def test_rgl_golden_fixtures(name):
    """tests/golden/<name>_rgb.bsdf (a synthetic file with the RGL field names, written by make_golden.py) through
    mrl_material_load_rgl against the committed outputs of oracle/rgl_oracle.c: eval / pdf at 1e-6; a sampled direction may differ from
    the fixture's by a Float ulp (which moves its pdf and weight by more than 1e-6 near the specular peak), so pdf / weight are
    held against the fixture only where the direction is bit-identical."""
    import os
    from mitsuba_customization_amd import host
    here = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(here, name + ".npz"))
    import torch
    with host.MerlHip(0) as g:
        mid = g.load_rgl(os.path.join(here, str(z["bsdf_file"])))
        wi, wo, u = (torch.from_numpy(z[k]).cuda() for k in ("wi", "wo", "u"))
        rgb, pdf, wo2, pdf2, w = (t.cpu().numpy() for t in g.eval_sample(wi, wo, u, material=mid))
    _close(rgb, z["rgb"], "eval"); _close(pdf, z["pdf"], "pdf")
    assert float(np.abs(rgb[3]).max()) == 0.0 and pdf[3] == 0.0                          # wi below the horizon
    assert np.count_nonzero((pdf2 > 0) != (z["pdf2"] > 0)) <= 1
    assert float(np.abs(wo2 - z["wo2"])[(pdf2 > 0) & (z["pdf2"] > 0)].max()) < 5e-7
    same = (wo2.view(np.int32) == z["wo2"].view(np.int32)).all(axis=1) & (pdf2 > 0)
    assert same.mean() > 0.6
    _close(pdf2[same], z["pdf2"][same], "sample pdf"); _close(w[same], z["weight"][same], "weight")


def _near_mirror_pairs(rng, n):
    """Pairs around the specular configuration wo = (-wi.x, -wi.y, wi.z): the exact mirror direction in Float, and that direction
    moved by 1 .. 64 Float ulps in one or two components; incidences from the normal to grazing (cos theta_i down to 1e-4)."""
    z = np.concatenate([rng.uniform(0.05, 1.0, n // 2), 10.0 ** rng.uniform(-4, -1.3, n - n // 2)])
    phi = rng.uniform(0, 2 * np.pi, n)
    r = np.sqrt(np.maximum(1.0 - z * z, 0.0))
    wi = np.stack([r * np.cos(phi), r * np.sin(phi), z], 1).astype(np.float32)
    wo = wi * np.array([-1, -1, 1], np.float32)
    k = rng.choice([0, 1, 2, 5, 17, 64], n)
    axis = rng.integers(0, 3, n)
    both = rng.random(n) < 0.3
    bits = wo.view(np.int32).copy()
    rows = np.arange(n)
    bits[rows, axis] += k * rng.choice([-1, 1], n)
    bits[rows[both], (axis[both] + 1) % 3] += k[both]
    wo = bits.view(np.float32)
    wo[:, 2] = np.abs(wo[:, 2])
    return wi, np.ascontiguousarray(wo)


@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[3]], ids=["isotropic", "anisotropic", "half_azimuth"])
def test_near_mirror_pairs_lie_in_the_oracles_rounding_range(case):
    """The ill-conditioned family, bounded: every value of every near-mirror pair is within 1e-6 of the oracle or inside the range the
    oracle itself spans over the rounding box of its half vector; and the well-conditioned majority (transverse half-vector length
    above 1e-6) meets 1e-6 without exception."""
    import torch
    from mitsuba_customization_amd import host, synth
    from oracle.binding import OracleRgl
    fields = synth.make_rgl_fields(**case)
    orc = OracleRgl(fields)
    wi, wo = _near_mirror_pairs(np.random.default_rng(4242 + case["seed"]), 20000)
    with host.MerlHip(0) as g:
        mid = g.upload_rgl(fields)
        rgb, pdf = (t.cpu().numpy() for t in g.eval_pdf(torch.from_numpy(wi).cuda(), torch.from_numpy(wo).cuda(), material=mid))
    o_rgb, o_pdf = orc.eval_pdf(wi, wo)
    assert float(o_pdf.max()) > 0 and np.isfinite(rgb).all() and np.isfinite(pdf).all()
    a = wi.astype(np.float64); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = wo.astype(np.float64); b /= np.linalg.norm(b, axis=1, keepdims=True)
    m = a + b
    t = np.hypot(m[:, 0], m[:, 1]) / np.linalg.norm(m, axis=1)
    well = t > 1e-6
    assert 0.2 < well.mean() < 0.98
    _close(rgb[well], o_rgb[well], "eval"); _close(pdf[well], o_pdf[well], "pdf")
    n_ill = _close(rgb[~well], o_rgb[~well], "eval", orc, wi[~well], wo[~well], max_ill=int((~well).sum()))
    n_ill += _close(pdf[~well], o_pdf[~well], "pdf", orc, wi[~well], wo[~well], max_ill=int((~well).sum()))
    print(f"near-mirror pairs: {int((~well).sum())} with a transverse half vector below 1e-6, {n_ill} values beyond 1e-6, all inside the oracle's rounding range")


def test_parity_soak_measure_at_a_few_rounds():
    """tools/fuzz_parity_rgl.py's measure (random file shapes and entry points; strict 1e-6 relative, the conditioning range for the
    rest) at 8 rounds x 2^15 units: nothing outside."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_parity_rgl", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "fuzz_parity_rgl.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    res = mod.soak(8, 1 << 15)
    assert res["outside_the_oracles_rounding_range"] == {k: 0 for k in res["outside_the_oracles_rounding_range"]}, res
    assert res["sampled_above_horizon_mismatches"] <= 2 and res["beyond_1e-6 (direction: 5e-7 absolute)"]["direction_abs"] == 0, res
    assert res["beyond_1e-6 (direction: 5e-7 absolute)"]["eval"] == 0 and res["beyond_1e-6 (direction: 5e-7 absolute)"]["pdf"] == 0, res


def test_marginal_rows_in_lds_when_the_conditional_integrals_do_not_fit():
    """An anisotropic file whose search tables exceed a CU's LDS (30 slices x 39 x 39 cells: 740 KB) but whose marginal rows fit
    (25 KB): the sample modes run k_rgl_lds<.., MARG_ONLY> — row searches from LDS, conditional integrals from memory — and answer
    with the bits of the all-memory kernel; spot-checked against the oracle."""
    import torch
    from mitsuba_customization_amd import host, synth
    from oracle.binding import OracleRgl
    fields = synth.make_rgl_fields(seed=71, n_phi=6, n_theta=5, res=40, res_ndf=16, res_sigma=8)
    n = 1 << 16
    with host.MerlHip(0) as g:
        mid = g.upload_rgl(fields)
        wi, wo, u = g.generate_pairs(71, 0, n)
        lds = [t.clone() for t in g.eval_sample(wi, wo, u, material=mid)]
        s_lds = [t.clone() for t in g.sample(wi, u, material=mid)]
        q = torch.arange(0, n, 2, device="cuda", dtype=torch.int32)
        cnt = torch.tensor([q.numel()], device="cuda", dtype=torch.int32)
        q_lds = [t.clone() for t in g.eval_sample_queue(wi, wo, u, q, cnt, material=mid)]
        g.set_option(host.OPT_RGL_SEARCH, 1)
        mem = g.eval_sample(wi, wo, u, material=mid)
        for a, b in zip(lds, mem):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        for a, b in zip(s_lds, g.sample(wi, u, material=mid)):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        for a, b in zip(q_lds, mem):
            assert torch.equal(a[0::2].view(torch.int32), b[0::2].view(torch.int32))
    orc = OracleRgl(fields)
    k = 4096
    hwi, hu = wi[:k].cpu().numpy(), u[:k].cpu().numpy()
    wo2, pdf2, w = (t[:k].cpu().numpy() for t in lds[2:])
    o_wo2, o_pdf2, _ = orc.sample(hwi, hu)
    both = (pdf2 > 0) & (o_pdf2 > 0)
    assert np.count_nonzero((pdf2 > 0) != (o_pdf2 > 0)) <= 1 and float(np.abs(wo2[both] - o_wo2[both]).max()) < 5e-7
    c_rgb, c_pdf = orc.eval_pdf(hwi[both], wo2[both])
    _close(pdf2[both], c_pdf, "sample pdf", orc, hwi[both], wo2[both], max_ill=1)
    _close(w[both], c_rgb / c_pdf[:, None], "weight", orc, hwi[both], wo2[both], max_ill=1)
