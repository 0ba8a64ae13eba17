"""The ABI's own device-memory helpers (for hosts without an allocator of their own), exercised through ctypes
with no torch tensor involved: alloc -> copy in -> batch on device pointers (own stream) -> copy out."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_device_alloc_copy_roundtrip_and_batch(oracle, tables):
    from mitsuba_customization_amd import host
    tab = tables("ggx_tab", 0)
    n = 10_000
    wi, wo, u = oracle.generate_pairs(0x5EED, 9, n)
    with host.MerlHip(0) as g:
        L, ctx = g._lib, g._ctx
        g.use_own_stream()
        mid = g.upload_merl(tab)
        ptr = {}
        for name, nbytes in (("wi", 12 * n), ("wo", 12 * n), ("rgb", 12 * n)):
            p = C.c_void_p()
            assert L.mrl_device_alloc(ctx, nbytes, C.byref(p)) == 0 and p.value
            ptr[name] = p
        assert L.mrl_copy_to_device(ctx, ptr["wi"], wi.ctypes.data, wi.nbytes) == 0
        assert L.mrl_copy_to_device(ctx, ptr["wo"], wo.ctypes.data, wo.nbytes) == 0
        assert L.mrl_timer_start(ctx) == 0
        assert L.mrl_eval_batch(ctx, ptr["wi"], ptr["wo"], None, mid, n, ptr["rgb"]) == 0
        ms = C.c_float()
        assert L.mrl_timer_stop(ctx, C.byref(ms)) == 0 and ms.value > 0
        out = np.empty((n, 3), np.float32)
        assert L.mrl_copy_to_host(ctx, out.ctypes.data, ptr["rgb"], out.nbytes) == 0
        back = np.empty_like(wi)
        assert L.mrl_copy_to_host(ctx, back.ctypes.data, ptr["wi"], back.nbytes) == 0 and np.array_equal(back, wi)
        for p in ptr.values():
            assert L.mrl_device_free(ctx, p) == 0
        assert L.mrl_device_free(ctx, None) == 0
        kind, dims = g.material_info(mid)
        assert kind == host.KIND_MERL and dims == (90, 90, 180) and g.material_count() == 1
        with pytest.raises(host.MerlHipError):
            g.material_info(5)
    want = oracle.OracleTable(tab).eval(wi, wo)
    assert (np.abs(out.astype(np.float64) - want) <= 1e-6 * np.abs(want) + 1e-30).all()


def test_customized_table_file_with_f32_payload(tmp_path):
    """A customized_measurement file may store f32 values: same header, payload width told by the file length."""
    import struct
    import torch
    from mitsuba_customization_amd import host, synth
    dims = (20, 16, 30)
    tab32 = synth.make_table("ggx_tab", seed=2, dims=dims).astype(np.float32)
    path = tmp_path / "custom_f32.binary"
    with open(path, "wb") as f:
        f.write(struct.pack("<3i", *dims))
        f.write(tab32.tobytes())
    with host.MerlHip(0) as g:
        a = g.load_table(str(path), scale=(1.0, 0.5, 2.0))
        b = g.upload_table(tab32.astype(np.float64), scale=(1.0, 0.5, 2.0))
        assert g.material_info(a) == g.material_info(b) == (host.KIND_TABLE, dims)
        wi, wo, u = g.generate_pairs(3, 0, 50000)
        for x, y in zip(g.eval_sample(wi, wo, u, material=a), g.eval_sample(wi, wo, u, material=b)):
            assert torch.equal(x.view(torch.int32), y.view(torch.int32))
        # a length that is neither payload width is a truncated file
        with open(path, "ab") as f:
            f.write(b"\\0" * 8)
        with pytest.raises(host.MerlHipError):
            g.load_table(str(path))


def test_pipelined_host_path_equals_staged_and_device_paths(oracle, tables):
    """Plain numpy arrays in, numpy arrays out: the pipelined path (copy threads + pinned double buffers + zero-copy
    kernel, MRL_OPT_HOST_THREADS > 0) against the staged hipMemcpy path (0) and the device-pointer path — bit for bit,
    over several chunks with a ragged tail, every entry point, a mixed batch and an n-channel table."""
    import torch
    from mitsuba_customization_amd import host, synth
    n = (1 << 21) + 12_345                                     # 3 pipeline chunks of 2^20, the last one ragged
    wi, wo, u = oracle.generate_pairs(0x5EED, 31, n)
    mat = oracle.generate_materials(0x5EED, 31, n, 3).astype(np.int32)
    with host.MerlHip(0) as g:
        ids = [g.upload_merl(tables("ggx_tab", s)) for s in range(3)]
        nch = g.upload_table_nch(synth.make_table_nch("spectral", 6, 2, (16, 12, 20)))
        assert g.get_option(host.OPT_HOST_THREADS) == 4
        dwi, dwo, du, dmat = (torch.from_numpy(a).cuda() for a in (wi, wo, u, mat))
        dev = [t.cpu().numpy() for t in g.eval_sample(dwi, dwo, du, mat=dmat)]
        dev_nch = [t.cpu().numpy() for t in g.eval_sample_nch(dwi, dwo, du, 6, material=nch)]
        results = {}
        for threads in (4, 1, 0, 7):
            g.set_option(host.OPT_HOST_THREADS, threads)
            results[threads] = {
                "fused": g.eval_sample(wi, wo, u, mat=mat),
                "eval": g.eval(wi, wo, material=ids[1]),
                "pdf": g.pdf(wi, wo, material=ids[1]),
                "sample": g.sample(wi, u, mat=mat),
                "eval_pdf": g.eval_pdf(wi, wo, material=ids[2]),
                "nch": g.eval_sample_nch(wi, wo, u, 6, material=nch),
            }
        with pytest.raises(host.MerlHipError):
            g.set_option(host.OPT_HOST_THREADS, 65)
    for threads, r in results.items():
        for a, b in zip(r["fused"], dev):
            assert np.array_equal(a.view(np.int32), b.view(np.int32)), threads
        for a, b in zip(r["nch"], dev_nch):
            assert np.array_equal(a.view(np.int32), b.view(np.int32)), threads
        for key in ("eval", "pdf"):
            assert np.array_equal(r[key], results[0][key]), (threads, key)
        for key in ("sample", "eval_pdf"):
            assert all(np.array_equal(a, b) for a, b in zip(r[key], results[0][key])), (threads, key)
    want = oracle.eval_sample_multi([oracle.OracleTable(tables("ggx_tab", s)) for s in range(3)], wi, wo, u, mat)
    assert (np.abs(results[4]["fused"][0].astype(np.float64) - want[0]) <= 1e-6 * np.abs(want[0]) + 1e-30).all()
    assert np.array_equal(results[4]["fused"][2], want[2])


def test_one_context_called_from_many_threads(oracle, tables):
    """The context's internal lock: 8 host threads hammer ONE context with device-pointer calls, host-array calls,
    queue calls and upload / release cycles (ctypes drops the GIL inside every call); every result equals the
    single-threaded one bit for bit."""
    import threading
    import torch
    from mitsuba_customization_amd import host
    n = 200_000
    wi, wo, u = oracle.generate_pairs(0x5EED, 77, n)
    with host.MerlHip(0) as g:
        g.use_own_stream()
        keep = g.upload_merl(tables("ggx_tab", 0))
        dwi, dwo, du = (torch.from_numpy(a).cuda() for a in (wi, wo, u))
        torch.cuda.synchronize()
        L, ctx = g._lib, g._ctx
        ref = [t.cpu().numpy() for t in g.eval_sample(dwi, dwo, du, material=keep)]
        g.use_own_stream()                                     # eval_sample() switched to torch's stream: back to the context's own
        errors = []

        def addr(t):
            return t.data_ptr()

        def worker(k):
            try:
                for it in range(12):
                    kind = (k + it) % 4
                    if kind == 0:                              # device pointers, raw ABI (no stream switching from this thread)
                        outs = [torch.empty((n, 3), device="cuda"), torch.empty((n,), device="cuda"), torch.empty((n, 3), device="cuda"),
                                torch.empty((n,), device="cuda"), torch.empty((n, 3), device="cuda")]
                        rc = L.mrl_eval_sample_batch(ctx, addr(dwi), addr(dwo), addr(du), None, keep, n, *[addr(o) for o in outs])
                        rc = rc or L.mrl_synchronize(ctx)
                        if rc != 0 or any(not np.array_equal(o.cpu().numpy(), r) for o, r in zip(outs, ref)):
                            errors.append(("device", k, it, rc))
                    elif kind == 1:                            # host arrays: the pipelined path, lock held for the call
                        outs = [np.empty((n, 3), np.float32), np.empty((n,), np.float32), np.empty((n, 3), np.float32),
                                np.empty((n,), np.float32), np.empty((n, 3), np.float32)]
                        rc = L.mrl_eval_sample_batch(ctx, wi.ctypes.data, wo.ctypes.data, u.ctypes.data, None, keep, n, *[o.ctypes.data for o in outs])
                        if rc != 0 or any(not np.array_equal(o, r) for o, r in zip(outs, ref)):
                            errors.append(("host", k, it, rc))
                    elif kind == 2:                            # eval only on host arrays
                        out = np.empty((n, 3), np.float32)
                        rc = L.mrl_eval_batch(ctx, wi.ctypes.data, wo.ctypes.data, None, keep, n, out.ctypes.data)
                        if rc != 0 or not np.array_equal(out, ref[0]):
                            errors.append(("eval", k, it, rc))
                    else:                                      # material churn beside the evaluations
                        mid = C.c_int()
                        tab = np.ascontiguousarray(tables("noise", 3, (8, 8, 16)), np.float64)
                        rc = L.mrl_material_upload_table(ctx, tab.ctypes.data, (C.c_int * 3)(8, 8, 16), (C.c_double * 3)(1, 1, 1), C.byref(mid))
                        rc = rc or L.mrl_material_release(ctx, mid.value)
                        if rc != 0 or mid.value == keep:
                            errors.append(("churn", k, it, rc))
            except Exception as e:                             # pragma: no cover
                errors.append(("exception", k, repr(e)))

        threads = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors[:5]
        assert g.material_count() >= 1 and g.material_info(keep)[0] == host.KIND_MERL


def test_block_maps_give_identical_results(tables):
    """MRL_OPT_BLOCK_MAP changes which workgroup evaluates which tile, never a value: whole-array and queue calls, single and
    mixed materials, sizes around the grid's edges (fewer tiles than blocks, a ragged last tile)."""
    import torch
    from mitsuba_customization_amd import host
    with host.MerlHip(0) as g:
        ids = [g.upload_merl(tables("ggx_tab", s)) for s in range(2)]
        for n in (1, 63, 257, 2049, 131_072 + 5, 3_000_001):
            wi, wo, u = g.generate_pairs(0x5EED, 9, n)
            mat = g.generate_materials(0x5EED, 9, n, 2)
            q = torch.arange(0, n, 3, device=wi.device, dtype=torch.int32)
            cnt = torch.tensor([q.numel()], device=wi.device, dtype=torch.int32)
            res = {}
            for bm in (0, 1):
                g.set_option(host.OPT_BLOCK_MAP, bm)
                res[bm] = [t.clone() for t in g.eval_sample(wi, wo, u, material=ids[0])] + [t.clone() for t in g.eval_sample(wi, wo, u, mat=mat)] + \
                          [t.clone() for t in g.eval_sample_queue(wi, wo, u, q, cnt, mat=mat)] + [g.eval(wi, wo, material=ids[1]).clone()]
            assert all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(res[0], res[1])), n


def test_reserved_compute_units_do_not_change_results():
    """MRL_OPT_RESERVED_CUS: the batch kernels run on a CU-masked stream with grids sized for the remaining CUs (what leaves room
    for RCCL's send / receive kernels beside a persistent grid); every entry point returns the same bits, the option validates."""
    import torch
    from mitsuba_customization_amd import host, synth
    with host.MerlHip(0) as g:
        tab = g.upload_merl(synth.make_table("ggx_tab", 2))
        ggx = g.ggx(0.3, (1.5, 1.5, 1.5), (3.0, 3.0, 3.0))
        rgl = g.upload_rgl(synth.make_rgl_fields(seed=3, n_phi=1, n_theta=4, res=8))
        n = 1 << 20
        wi, wo, u = g.generate_pairs(11, 0, n)
        ids = torch.tensor([tab, ggx, rgl], device="cuda", dtype=torch.int32)
        mat = ids[torch.arange(n, device="cuda") % 3]
        g.synchronize()
        want = [[t.clone() for t in g.eval_sample(wi, wo, u, material=m)] for m in (tab, ggx, rgl)] + [[t.clone() for t in g.eval_sample(wi, wo, u, mat=mat)]]
        g.synchronize()
        for k in (8, 16, 0):
            g.set_option(host.OPT_RESERVED_CUS, k)
            assert g.get_option(host.OPT_RESERVED_CUS) == k
            got = [g.eval_sample(wi, wo, u, material=m) for m in (tab, ggx, rgl)] + [g.eval_sample(wi, wo, u, mat=mat)]
            g.synchronize()
            for a, b in zip(got, want):
                for x, y in zip(a, b):
                    assert torch.equal(x.view(torch.int32), y.view(torch.int32)), k
        for bad in (-1, 17, 1000):
            with pytest.raises(host.MerlHipError):
                g.set_option(host.OPT_RESERVED_CUS, bad)
