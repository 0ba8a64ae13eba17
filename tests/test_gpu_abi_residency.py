"""Material lifetime and memory accounting of the C ABI: mrl_material_release (tombstones, slot reuse),
mrl_memory_info, the MRL_OPT_MEMORY_LIMIT_MB budget and MRL_ERR_OOM, option range checks."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MERL_BRICK_BYTES = 90 * 90 * 180 * 128
MERL_ROWS_BYTES = 91 * 91 * 181 * 16


def test_memory_info_counts_resident_tables(tables):
    from mitsuba_customization_amd import host
    with host.MerlHip(0) as g:
        assert g.memory_info()["table_bytes"] == 0
        a = g.upload_merl(tables("ggx_tab", 0))
        one = g.memory_info()
        assert MERL_BRICK_BYTES <= one["table_bytes"] <= MERL_BRICK_BYTES + 65536     # + the sampling marginal (272 doubles) + the conditional table (32 x 181 doubles)
        assert 0 < one["device_free"] < one["device_total"]
        g.ggx(0.1, (1, 1, 1), (2, 2, 2))
        assert g.memory_info()["table_bytes"] == one["table_bytes"]                    # analytic materials hold no table
        b = g.upload_table(tables("noise", 3, (8, 8, 16)))
        small = 8 * 8 * 16 * 128 + (3 * 8 + 2) * 8 + 32 * (2 * 8 + 1) * 8       # bricks + row marginal + 32 conditional rows of (2 n_th + 1) doubles
        assert g.memory_info()["table_bytes"] == one["table_bytes"] + small
        g.release_material(a)
        assert g.memory_info()["table_bytes"] == small
        g.release_material(b)
        assert g.memory_info()["table_bytes"] == 0
    with host.MerlHip(0) as g:
        g.set_option(host.OPT_TABLE_LAYOUT, host.LAYOUT_ROWS)
        g.upload_merl(tables("ggx_tab", 0))
        assert MERL_ROWS_BYTES <= g.memory_info()["table_bytes"] <= MERL_ROWS_BYTES + 65536


def test_release_gives_memory_back_and_tombstones_render_zero(oracle, tables):
    import torch
    from mitsuba_customization_amd import host
    n = 4096
    with host.MerlHip(0) as g:
        keep = g.upload_merl(tables("ggx_tab", 1))
        free0 = g.memory_info()["device_free"]
        gone = g.upload_merl(tables("ggx_tab", 0))
        assert free0 - g.memory_info()["device_free"] >= MERL_BRICK_BYTES - (64 << 20)
        wi, wo, u = g.generate_pairs(0x5EED, 0, n)
        mat = torch.where(torch.arange(n, device=wi.device) % 2 == 0, keep, gone).to(torch.int32)
        before = [t.clone() for t in g.eval_sample(wi, wo, u, mat=mat)]
        g.release_material(gone)
        assert abs(g.memory_info()["device_free"] - free0) <= (64 << 20)               # the table left HBM
        # the released id: single-material calls reject it, batches render it as an unknown id (zeros)
        with pytest.raises(host.MerlHipError) as e:
            g.eval(wi, wo, material=gone)
        assert e.value.status == host.ERR_MATERIAL
        with pytest.raises(host.MerlHipError):
            g.material_info(gone)
        with pytest.raises(host.MerlHipError):
            g.release_material(gone)                                                  # double release
        assert g.material_count() == 2                                                # slots, not live materials
        for variant in (0, 1, 2, 3):
            g.set_option(host.OPT_KERNEL, variant)
            after = g.eval_sample(wi, wo, u, mat=mat)
            for b, a in zip(before, after):
                assert torch.equal(a[0::2].view(torch.int32), b[0::2].view(torch.int32)), variant   # the kept material is untouched
                assert float(a[1::2].abs().max()) == 0.0, variant                                   # the tombstone renders zeros
        g.set_option(host.OPT_KERNEL, 3)
        # queue entry points see the tombstone the same way
        q = torch.arange(n, device=wi.device, dtype=torch.int32)
        cnt = torch.tensor([n], device=wi.device, dtype=torch.int32)
        outq = g.eval_sample_queue(wi, wo, u, q, cnt, mat=mat)
        assert float(outq[0][1::2].abs().max()) == 0.0 and torch.equal(outq[0][0::2], before[0][0::2])
        # the slot is reused by the next upload (lowest free slot first), and works
        again = g.upload_merl(tables("ggx_tab", 0))
        assert again == gone and g.material_count() == 2
        now = g.eval_sample(wi, wo, u, mat=mat)
        for b, a in zip(before, now):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        # releasing every material leaves a context that still answers (with zeros) and accepts new uploads
        g.release_material(keep); g.release_material(again)
        z = g.eval_sample(wi, wo, u, mat=mat)
        assert all(float(t.abs().max()) == 0.0 for t in z)
        assert g.upload_merl(tables("ggx_tab", 1)) == 0


def test_upload_release_cycles_keep_free_memory_flat(tables):
    from mitsuba_customization_amd import host
    tab = tables("ggx_tab", 0)
    with host.MerlHip(0) as g:
        g.upload_merl(tables("ggx_tab", 1))
        base = g.memory_info()["device_free"]
        for _ in range(50):
            mid = g.upload_merl(tab)
            assert mid == 1
            g.release_material(mid)
            assert abs(g.memory_info()["device_free"] - base) <= (64 << 20)
        assert g.memory_info()["table_bytes"] <= MERL_BRICK_BYTES + 65536


def test_memory_budget_and_oom(tables):
    from mitsuba_customization_amd import host
    tab = tables("ggx_tab", 0)
    with host.MerlHip(0) as g:
        L, ctx = g._lib, g._ctx
        g.set_option(host.OPT_MEMORY_LIMIT_MB, 400)                 # two brick tables of 178 MiB fit, a third does not
        assert g.get_option(host.OPT_MEMORY_LIMIT_MB) == 400
        a = g.upload_merl(tab); b = g.upload_merl(tab)
        used = g.memory_info()["table_bytes"]
        free_before = g.memory_info()["device_free"]
        with pytest.raises(host.MerlHipError) as e:
            g.upload_merl(tab)
        assert e.value.status == host.ERR_OOM and "budget" in str(e.value)
        assert g.memory_info()["table_bytes"] == used and g.material_count() == 2      # the context is as it was
        assert abs(g.memory_info()["device_free"] - free_before) <= (64 << 20)
        g.release_material(a)
        assert g.upload_merl(tab) == a                                                 # room again
        g.set_option(host.OPT_MEMORY_LIMIT_MB, 0)
        # the device itself: an allocation larger than the card is MRL_ERR_OOM, not a crash
        p = C.c_void_p()
        assert L.mrl_device_alloc(ctx, 1 << 42, C.byref(p)) == host.ERR_OOM
        wi, wo, u = g.generate_pairs(1, 0, 1000)
        assert float(g.eval(wi, wo, material=b).abs().max()) > 0                       # still alive


def test_option_ranges():
    from mitsuba_customization_amd import host
    with host.MerlHip(0) as g:
        L, ctx = g._lib, g._ctx
        for v in (0, 1, 2, 3, 4):
            assert L.mrl_set_option(ctx, host.OPT_KERNEL, v) == 0
        for v in (-1, 5, 99):
            assert L.mrl_set_option(ctx, host.OPT_KERNEL, v) == host.ERR_INVALID
        assert g.get_option(host.OPT_KERNEL) == 4
        assert L.mrl_set_option(ctx, host.OPT_MEMORY_LIMIT_MB, -5) == host.ERR_INVALID
        assert L.mrl_set_option(ctx, host.OPT_BLOCK_MAP, 2) == host.ERR_INVALID and L.mrl_set_option(ctx, host.OPT_BLOCK_MAP, 1) == 0
        assert L.mrl_material_release(ctx, 0) == host.ERR_MATERIAL and L.mrl_material_release(None, 0) == host.ERR_INVALID
        assert L.mrl_memory_info(None, None, None, None, None) == host.ERR_INVALID
        assert L.mrl_memory_info(ctx, None, None, None, None) == 0
