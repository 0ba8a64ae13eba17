"""n-channel tables on the GPU (mrl_*_nch) against the oracle's n-channel restatement and the
committed fixtures.  Tolerances as for RGB: sampled direction / pdf (cosine sampling) bit-identical, values and weights
|gpu - oracle| <= 1e-6 |oracle| for every value.  PARITY UNPINNED (the reference's customized_measurement format is
unknown): the oracle is this repo's own restatement."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _close(got, want, tol=1e-6):
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    return bool((np.abs(got - want) <= tol * np.abs(want) + 1e-30).all())


def _check(gpu_out, want, bitexact_dirs=True):
    val, pdf, wo2, pdf2, w = [t.cpu().numpy() if hasattr(t, "cpu") else t for t in gpu_out]
    assert _close(val, want[0]), "values"
    if bitexact_dirs:
        assert np.array_equal(pdf, want[1]) and np.array_equal(wo2, want[2]) and np.array_equal(pdf2, want[3])
    assert _close(w, want[4]), "weight"


@pytest.mark.parametrize("n_ch,kind,dims", [(1, "noise", (24, 20, 36)), (2, "noise", (16, 12, 20)), (4, "spectral", (30, 24, 40)),
                                             (5, "noise", (10, 12, 14)), (16, "spectral", (20, 16, 24)), (32, "noise", (6, 5, 8))])
@pytest.mark.parametrize("lookup,node,disk", [(1, 0, 0), (1, 1, 1), (0, 0, 0)])
def test_nch_eval_sample_matches_oracle(oracle, n_ch, kind, dims, lookup, node, disk):
    from mitsuba_customization_amd import host, synth
    tab = synth.make_table_nch(kind, n_ch, 7, dims)
    scale = [0.5 + 0.25 * c for c in range(n_ch)]
    n = 20_011
    with host.MerlHip(0) as g:
        g.set_option(host.OPT_LOOKUP, lookup); g.set_option(host.OPT_NODE, node); g.set_option(host.OPT_DISK_MAP, disk)
        mid = g.upload_table_nch(tab, scale)
        assert g.material_channels(mid) == n_ch and g.material_info(mid) == (host.KIND_TABLE_NCH, dims)
        wi, wo, u = g.generate_pairs(0x5EED, 1234, n)
        wi[3, 2] = -wi[3, 2]; wo[5, 2] = -wo[5, 2]                       # below-horizon guards
        fused = g.eval_sample_nch(wi, wo, u, n_ch, material=mid)
        ev = g.eval_nch(wi, wo, n_ch, material=mid)
        ep = g.eval_pdf_nch(wi, wo, n_ch, material=mid)
        sm = g.sample_nch(wi, u, n_ch, material=mid)
        pdf_only = g.pdf(wi, wo, material=mid)                           # the channel-free call serves n-channel tables too
        hin = [t.cpu().numpy() for t in (wi, wo, u)]
    want = oracle.eval_sample_nch([oracle.OracleTableNch(tab, scale)], *hin, None, oracle.make_opts(lookup, node, disk))
    if lookup:
        _check(fused, want)
    else:                                                               # nearest: a coordinate on an exact bin edge may flip
        ok = np.abs(fused[0].cpu().numpy().astype(np.float64) - want[0]) <= 1e-6 * np.abs(want[0]) + 1e-30
        assert (~ok.all(axis=1)).sum() <= 1                             # measured: none
        assert np.array_equal(fused[2].cpu().numpy(), want[2]) and np.array_equal(fused[3].cpu().numpy(), want[3])
    import torch
    for a, b in ((ev, fused[0]), (ep[0], fused[0]), (ep[1], fused[1]), (sm[0], fused[2]), (sm[1], fused[3]), (sm[2], fused[4]), (pdf_only, fused[1])):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))    # every entry point is the same arithmetic


@pytest.mark.parametrize("name", ["nch_c1_noise", "nch_c4_spectral", "nch_c16_spectral_table_sampling"])
def test_nch_golden_fixtures(name):
    from mitsuba_customization_amd import host, synth
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    n_ch, dims = int(z["n_ch"]), tuple(int(d) for d in z["dims"])
    tab = synth.make_table_nch(str(z["table_kind"]), n_ch, int(z["table_seed"]), dims)
    with host.MerlHip(0) as g:
        g.set_option(host.OPT_SAMPLING, int(z["sampling"]))
        mid = g.upload_table_nch(tab, z["scale"].tolist())
        out = g.eval_sample_nch(z["wi"], z["wo"], z["u"], n_ch, material=mid)       # host arrays: the staged path
    want = (z["rgb"], z["pdf"], z["wo2"], z["pdf2"], z["weight"])
    if int(z["sampling"]) == 0:
        _check(out, want)
    else:
        # table sampling: the half-vector branch computes its direction in f64 and rounds once -> within one f32 ulp;
        # a direction one ulp across a theta_h bin edge changes bins (documented for the RGB path as well)
        assert _close(out[0], want[0])
        assert np.abs(out[2].astype(np.float64) - want[2]).max() <= 1.2e-7
        for got, ref in ((out[1], want[1]), (out[3], want[3]), (out[4], want[4])):
            ok = np.abs(got.astype(np.float64) - ref) <= 3e-6 * np.abs(ref) + 1e-30
            assert ok.all() if got is out[1] else ok.mean() > 0.9995
        # and EVERY unit against the oracle evaluated at the direction the device returned
        from oracle import binding as ob
        T = ob.OracleTableNch(tab, z["scale"].tolist())
        at = ob.eval_sample_nch([T], z["wi"], out[2], z["u"], table_sampling=True, n_ch=n_ch)
        live = out[3] > 0
        assert np.array_equal(live, at[1] > 0)
        assert (np.abs(out[3][live].astype(np.float64) - at[1][live]) <= 2e-6 * at[1][live]).all()
        w_at = at[0][live].astype(np.float64) / at[1][live].astype(np.float64)[:, None]
        assert (np.abs(out[4][live].astype(np.float64) - w_at) <= 3e-6 * np.abs(w_at) + 1e-30).all()


def test_nch_mixed_batch_and_wrong_widths(oracle):
    """A batch over several 4-channel tables; ids of a 2-channel table, an RGB table, a GGX material, a released slot and
    out-of-range ids all render zeros; the RGB entry points in turn refuse / zero n-channel materials."""
    import torch
    from mitsuba_customization_amd import host, synth
    dims = (12, 10, 16)
    t4a, t4b = synth.make_table_nch("noise", 4, 1, dims), synth.make_table_nch("spectral", 4, 2, dims)
    t2 = synth.make_table_nch("noise", 2, 3, dims)
    rgb = synth.make_table("noise", 4, dims)
    n = 30_000
    with host.MerlHip(0) as g:
        a = g.upload_table_nch(t4a); b = g.upload_table_nch(t2); c = g.upload_table(rgb); d = g.ggx(0.2, (1, 1, 1), (2, 2, 2))
        e = g.upload_table_nch(t4b); dead = g.upload_table_nch(t4a); g.release_material(dead)
        assert (a, b, c, d, e, dead) == (0, 1, 2, 3, 4, 5)
        wi, wo, u = g.generate_pairs(0x5EED, 99, n)
        mat = (torch.arange(n, device=wi.device) % 8 - 1).to(torch.int32)          # -1 .. 6
        out = [t.cpu().numpy() for t in g.eval_sample_nch(wi, wo, u, 4, mat=mat)]
        hin = [t.cpu().numpy() for t in (wi, wo, u)]
        hm = mat.cpu().numpy()
        # oracle view: slots 0 and 4 are 4-channel tables, everything else is "not evaluable by this call"
        dummy = oracle.OracleTableNch(np.zeros((1, 1, 1, 1)))
        tabs = [oracle.OracleTableNch(t4a), dummy, dummy, dummy, oracle.OracleTableNch(t4b)]
        want = oracle.eval_sample_nch(tabs, *hin, hm, n_ch=4)
        _check(out, want)
        for k in (-1, 1, 2, 3, 5, 6):
            assert not out[0][hm == k].any() and not out[4][hm == k].any() and not out[1][hm == k].any()
        assert out[0][hm == 0].any() and out[0][hm == 4].any()
        # RGB calls: an n-channel single id is refused, n-channel ids inside a batch render zeros
        with pytest.raises(host.MerlHipError) as err:
            g.eval(wi, wo, material=a)
        assert err.value.status == host.ERR_MATERIAL and "nch" in str(err.value)
        rgb_out = g.eval_sample(wi, wo, u, mat=mat)
        hr = rgb_out[0].cpu().numpy()
        assert hr[hm == 2].any() and hr[hm == 3].any() and not hr[(hm != 2) & (hm != 3)].any()
        # an _nch call with the wrong width for a single id / unsupported widths
        with pytest.raises(host.MerlHipError):
            g.eval_nch(wi, wo, 4, material=b)
        with pytest.raises(host.MerlHipError):
            g.eval_nch(wi, wo, 33, material=a)
        with pytest.raises(host.MerlHipError):
            g.upload_table_nch(np.zeros((40, 2, 2, 2)))
        # three channels through the _nch calls = the RGB path
        v3 = g.eval_sample_nch(wi, wo, u, 3, material=c)
        r3 = g.eval_sample(wi, wo, u, material=c)
        assert all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in zip(v3, r3))
        assert g.upload_table_nch(np.abs(rgb)) == dead and g.material_info(dead)[0] == host.KIND_TABLE       # 3 planes -> RGB kind, freed slot


def test_four_channel_table_agrees_with_rgb_path(tables):
    """The first three channels of an (R, G, B, extra) table equal the RGB path's result on (R, G, B): same transform,
    same texels, same weights — only the brick layout differs."""
    import torch
    from mitsuba_customization_amd import host, synth
    dims = (20, 16, 24)
    rgb = synth.make_table("ggx_tab", 3, dims)
    four = np.concatenate([rgb, synth.make_table("noise", 9, dims)[:1]], axis=0)
    with host.MerlHip(0) as g:
        a = g.upload_table(rgb, scale=(0.5, 2.0, 1.25))
        b = g.upload_table_nch(four, (0.5, 2.0, 1.25, 1.0))
        wi, wo, u = g.generate_pairs(0x5EED, 5, 100_000)
        r = g.eval_sample(wi, wo, u, material=a)
        f = g.eval_sample_nch(wi, wo, u, 4, material=b)
        assert _close(f[0][:, :3].cpu().numpy(), r[0].cpu().numpy(), 5e-7) and _close(f[4][:, :3].cpu().numpy(), r[4].cpu().numpy(), 5e-7)     # the RGB path blends in packed Float (bound 3.6e-7), the n-channel path in f64
        assert torch.equal(f[2], r[2]) and torch.equal(f[1], r[1]) and torch.equal(f[3], r[3])


def test_nch_file_loaders(oracle, tmp_path):
    from mitsuba_customization_amd import host, synth
    dims, C = (8, 6, 10), 6
    tab = synth.make_table_nch("spectral", C, 5, dims)
    scale = [1.0 + 0.1 * c for c in range(C)]
    f64, f32 = str(tmp_path / "six.binary"), str(tmp_path / "six_f32.binary")
    synth.write_table_nch(f64, tab)
    synth.write_table_nch(f32, tab, dtype="<f4")
    wi, wo, u = oracle.generate_pairs(0x5EED, 8, 5000)
    with host.MerlHip(0) as g:
        a = g.load_table_nch(f64, C, scale)
        b = g.load_table_nch(f32, C, scale)
        up = g.upload_table_nch(tab, scale)
        ra, rb, ru = (g.eval_sample_nch(wi, wo, u, C, material=m) for m in (a, b, up))
        assert all(np.array_equal(x, y) for x, y in zip(ra, ru))
        _check(rb, oracle.eval_sample_nch([oracle.OracleTableNch(tab.astype(np.float32), scale)], wi, wo, u))
        with pytest.raises(host.MerlHipError) as e:                     # a file length that fits neither payload width
            g.load_table_nch(f64, C + 1)
        assert e.value.status == host.ERR_FORMAT
        # the tensor_file container: the committed 5-channel fixture and a 3-channel one (-> RGB kind)
        mid, ch = g.load_tensor_table(os.path.join(os.path.dirname(__file__), "golden", "tensor_table_c5.bsdf"))
        assert ch == 5 and g.material_info(mid) == (host.KIND_TABLE_NCH, (6, 5, 8))
        t5 = synth.make_table_nch("spectral", 5, 9, (6, 5, 8)).astype(np.float32)
        _check(g.eval_sample_nch(wi, wo, u, 5, material=mid),
               oracle.eval_sample_nch([oracle.OracleTableNch(t5, [1.0, 0.5, 2.0, 1.5, 0.25])], wi, wo, u))
        p3 = str(tmp_path / "rgb.bsdf")
        rgb = synth.make_table("ggx_tab", 1, (10, 8, 12))
        synth.write_tensor_file(p3, {"brdf": rgb, "other": np.zeros(3, np.float32)})
        mid3, ch3 = g.load_tensor_table(p3, "brdf")
        assert ch3 == 3 and g.material_info(mid3)[0] == host.KIND_TABLE
        want = oracle.eval_sample_multi([oracle.OracleTable(rgb, (1.0, 1.0, 1.0))], wi, wo, u, None)
        got = g.eval_sample(wi, wo, u, material=mid3)
        assert _close(got[0], want[0]) and np.array_equal(got[2], want[2])
        for bad_field in ("other", "missing"):                          # not a [C, h, d, p] float table / absent
            with pytest.raises(host.MerlHipError) as e:
                g.load_tensor_table(p3, bad_field)
            assert e.value.status == host.ERR_FORMAT


def test_nch_large_batch_tile_invariance():
    """2^24 units over a 16-channel table: results do not depend on where a unit sits in the batch."""
    import torch
    from mitsuba_customization_amd import host, synth
    C, n = 16, 1 << 24
    with host.MerlHip(0) as g:
        mid = g.upload_table_nch(synth.make_table_nch("spectral", C, 1, (30, 30, 60)))
        wi, wo, u = g.generate_pairs(0x5EED, 0, n)
        full = g.eval_sample_nch(wi, wo, u, C, material=mid)
        lo, hi = 5_000_003, 5_000_003 + 777_777
        part = g.eval_sample_nch(wi[lo:hi].contiguous(), wo[lo:hi].contiguous(), u[lo:hi].contiguous(), C, material=mid)
        assert all(torch.equal(a[lo:hi].view(torch.int32), b.view(torch.int32)) for a, b in zip(full, part))
        assert float(full[0].min()) >= 0.0 and float(full[0].max()) > 0.0 and bool(torch.isfinite(full[4]).all())


def test_rgb_batches_never_read_a_narrow_table_as_bricks():
    """ids of a MERL-sized ONE-channel table (32 B per cell, 47 MB) inside an RGB batch: the RGB kernels read 128-B
    bricks, so evaluating such an id against its own descriptor would run 140 MB past its allocation; they evaluate
    the context's safe 1x1x1 table instead and render zeros (all kernel variants, both batch and queue calls)."""
    import torch
    from mitsuba_customization_amd import host, synth
    n = 1 << 20
    with host.MerlHip(0) as g:
        rgb = g.upload_merl(synth.make_table("ggx_tab", 0))
        mono = g.upload_table_nch(np.abs(synth.make_table("ggx_tab", 1)[:1]))
        wi, wo, u = g.generate_pairs(0x5EED, 0, n)
        mat = torch.where(torch.arange(n, device=wi.device) % 2 == 0, rgb, mono).to(torch.int32)
        ref = g.eval_sample(wi, wo, u, material=rgb)
        for variant in (0, 1, 2, 3):
            g.set_option(host.OPT_KERNEL, variant)
            out = g.eval_sample(wi, wo, u, mat=mat)
            for a, b in zip(out, ref):
                assert float(a[1::2].abs().max()) == 0.0, variant
                assert torch.equal(a[0::2], b[0::2]) or variant == 0       # variant 0 is the generic-math kernel: compared below
            if variant == 0:
                assert torch.allclose(out[0][0::2], ref[0][0::2], rtol=1e-6, atol=0)
        g.set_option(host.OPT_KERNEL, 3)
        q = torch.arange(n, device=wi.device, dtype=torch.int32)
        cnt = torch.tensor([n], device=wi.device, dtype=torch.int32)
        outq = g.eval_sample_queue(wi, wo, u, q, cnt, mat=mat)
        assert float(outq[0][1::2].abs().max()) == 0.0 and torch.equal(outq[0][0::2], ref[0][0::2])


@pytest.mark.parametrize("n_ch", [1, 4, 6, 16, 32])
def test_nch_queue_calls_match_whole_array_calls(n_ch):
    """mrl_*_queue_nch: queued slots equal the whole-array results bit for bit, every other slot keeps its sentinel;
    ragged counts, a count above the capacity (clamped), an empty queue, a shuffled queue, mixed materials."""
    import torch
    from mitsuba_customization_amd import host, synth
    n = 50_007
    dims = (14, 12, 18)
    with host.MerlHip(0) as g:
        a = g.upload_table_nch(synth.make_table_nch("spectral", n_ch, 1, dims))
        b = g.upload_table_nch(synth.make_table_nch("noise", n_ch, 2, dims))
        other = g.upload_table_nch(synth.make_table_nch("noise", 2 if n_ch != 2 else 1, 3, dims))       # another width: zeros
        wi, wo, u = g.generate_pairs(0x5EED, 7, n)
        mat = torch.tensor([a, b, other], device=wi.device, dtype=torch.int32)[torch.arange(n, device=wi.device) % 3].contiguous()
        whole = g.eval_sample_nch(wi, wo, u, n_ch, mat=mat)
        perm = torch.randperm(n, device=wi.device, generator=torch.Generator(device=wi.device).manual_seed(5))
        for queue, count in ((torch.arange(0, n, 3, device=wi.device), None), (perm[: n // 2], None), (torch.arange(0, n, 2, device=wi.device), 1000),
                             (torch.arange(64, device=wi.device), 0), (torch.arange(100, device=wi.device), 10_000)):
            queue = queue.to(torch.int32).contiguous()
            k = queue.numel() if count is None else count
            cnt = torch.tensor([k], device=wi.device, dtype=torch.int32)
            live = queue[: min(k, queue.numel())].long()
            sentinel = -7.0
            out = (torch.full((n, n_ch), sentinel, device=wi.device), torch.full((n,), sentinel, device=wi.device), torch.full((n, 3), sentinel, device=wi.device),
                   torch.full((n,), sentinel, device=wi.device), torch.full((n, n_ch), sentinel, device=wi.device))
            g.eval_sample_queue_nch(wi, wo, u, queue, cnt, n_ch, mat=mat, out=out)
            untouched = torch.ones(n, dtype=torch.bool, device=wi.device)
            untouched[live] = False
            for got, ref in zip(out, whole):
                assert torch.equal(got[live].view(torch.int32), ref[live].view(torch.int32))
                assert bool((got[untouched] == sentinel).all())
        # the single-output calls, single material
        q = torch.arange(0, n, 5, device=wi.device, dtype=torch.int32)
        cnt = torch.tensor([q.numel()], device=wi.device, dtype=torch.int32)
        ev = g.eval_queue_nch(wi, wo, q, cnt, n_ch, material=b)
        ep = g.eval_pdf_queue_nch(wi, wo, q, cnt, n_ch, material=b)
        sm = g.sample_queue_nch(wi, u, q, cnt, n_ch, material=b)
        ref = g.eval_sample_nch(wi, wo, u, n_ch, material=b)
        ql = q.long()
        assert torch.equal(ev[ql], ref[0][ql]) and torch.equal(ep[0][ql], ref[0][ql]) and torch.equal(ep[1][ql], ref[1][ql])
        assert torch.equal(sm[0][ql], ref[2][ql]) and torch.equal(sm[1][ql], ref[3][ql]) and torch.equal(sm[2][ql], ref[4][ql])
        rest = ev.clone(); rest[ql] = 0.0
        assert float(rest.abs().max()) == 0.0                                        # unqueued slots of a zero-initialised output stay zero


@pytest.mark.parametrize("n_ch", [4, 9])
def test_nch_guards_unnormalised_and_nonfinite_inputs(oracle, n_ch):
    """Below-horizon, zero, unnormalised, huge, NaN and inf directions through the narrow (4) and the wide (9 channels)
    kernel: zeros where the oracle has zeros, NaN where it has NaN, equal values elsewhere — no trap, no hang."""
    from mitsuba_customization_amd import host, synth
    tab = synth.make_table_nch("spectral", n_ch, 4, (24, 20, 30))
    wi = np.array([[0, 0, 1], [0.6, 0, 0.8], [0.6, 0, -0.8], [0.6, 0, 0.8], [1, 0, 0], [0.3, 0.4, 0.5], [3, 4, 5], [3e-5, 4e-5, 1e-6],
                   [np.nan, 0, 1], [0, 0, np.inf], [0, 0, 0], [1e30, 0, 1e30], [0.1, 0.2, 0.97]], np.float32)
    wo = np.array([[0, 0, 1], [-0.6, 0, 0.8], [0.6, 0, 0.8], [0.6, 0, -0.8], [0, 0, 1], [0.9, 1.2, 1.5], [0.09, 0.12, 0.15], [0, 1, 1e-3],
                   [0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 1e30, 1e30], [np.nan, np.nan, np.nan]], np.float32)
    u = np.tile(np.array([[0.3, 0.7]], np.float32), (wi.shape[0], 1))
    u[1] = (0.0, 0.0); u[2] = (1.0, 0.5); u[3] = (0.5, 0.5)
    with host.MerlHip(0) as g:
        mid = g.upload_table_nch(tab)
        got = g.eval_sample_nch(wi, wo, u, n_ch, material=mid)          # host arrays
        import torch
        dev = [t.cpu().numpy() for t in g.eval_sample_nch(*(torch.from_numpy(a).cuda() for a in (wi, wo, u)), n_ch, material=mid)]
    want = oracle.eval_sample_nch([oracle.OracleTableNch(tab)], wi, wo, u)
    for a, b in zip(got, dev):
        assert np.array_equal(a, b, equal_nan=True)
    for k, (a, b) in enumerate(zip(got, want)):
        assert np.array_equal(np.isnan(a), np.isnan(b)), k
        fin = ~np.isnan(b)
        assert (np.abs(a[fin].astype(np.float64) - b[fin]) <= 2e-6 * np.abs(b[fin]) + 1e-30).all(), k
    for row in (2, 3, 4, 10):                                           # guards: everything of eval is zero
        assert not got[0][row].any() and got[1][row] == 0
    assert np.allclose(got[0][5] / 1.5, got[0][6] / 0.15, rtol=1e-6)    # f depends on the direction only


@pytest.mark.parametrize("n_ch", [2, 8])
def test_nch_table_sampling_in_a_mixed_batch(oracle, n_ch):
    """MRL_OPT_SAMPLING = 1 over a batch of three n-channel tables (each with its own row marginal) and an unknown id:
    cosine half bit-identical, half-vector half within one f32 ulp, pdf / weight to 3e-6 for >= 99.9 % (a direction one
    ulp across a theta_h bin edge changes bins) and for EVERY unit against the oracle at the returned direction, sample().pdf == pdf(wi, sample().wo) on the device."""
    import torch
    from mitsuba_customization_amd import host, synth
    dims = (30, 20, 36)
    tabs = [synth.make_table_nch("spectral", n_ch, s, dims) for s in (1, 2, 3)]
    n = 60_000
    with host.MerlHip(0) as g:
        g.set_option(host.OPT_SAMPLING, 1)
        ids = [g.upload_table_nch(t) for t in tabs]
        wi, wo, u = g.generate_pairs(0x5EED, 11, n)
        mat = (torch.arange(n, device=wi.device) % 4).to(torch.int32)             # 3 = unknown
        out = g.eval_sample_nch(wi, wo, u, n_ch, mat=mat)
        back = g.pdf(wi, out[2].contiguous(), mat=mat)                            # pdf of the sampled direction, same option
        assert torch.equal(back.view(torch.int32), out[3].view(torch.int32))
        got = [t.cpu().numpy() for t in out]
        hin = [t.cpu().numpy() for t in (wi, wo, u)]
        hm = mat.cpu().numpy()
    want = oracle.eval_sample_nch([oracle.OracleTableNch(t) for t in tabs], *hin, hm, table_sampling=True, n_ch=n_ch)
    assert _close(got[0], want[0])
    assert np.abs(got[2].astype(np.float64) - want[2]).max() <= 1.2e-7
    cosine_half = hin[2][:, 0] < 0.5
    assert np.array_equal(got[2][cosine_half], want[2][cosine_half])
    for k in (1, 3, 4):
        ok = np.abs(got[k].astype(np.float64) - want[k]) <= 3e-6 * np.abs(want[k]) + 1e-30
        assert ok.all() if k == 1 else ok.mean() > 0.999, k               # the pdf query has no sampled direction in it
    # EVERY unit against the oracle evaluated at the direction the device returned: pdf(wi, wo') and eval(wi, wo') / pdf
    at = oracle.eval_sample_nch([oracle.OracleTableNch(t) for t in tabs], hin[0], got[2], hin[2], hm, table_sampling=True, n_ch=n_ch)
    live = got[3] > 0
    assert np.array_equal(live, at[1] > 0)
    assert (np.abs(got[3][live].astype(np.float64) - at[1][live]) <= 2e-6 * at[1][live]).all()
    w_at = at[0][live].astype(np.float64) / at[1][live].astype(np.float64)[:, None]
    assert (np.abs(got[4][live].astype(np.float64) - w_at) <= 3e-6 * np.abs(w_at) + 1e-30).all()
    assert not got[0][hm == 3].any() and not got[4][hm == 3].any() and not got[3][hm == 3].any()
