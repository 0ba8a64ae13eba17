"""Argument validation of the C ABI on a live context: every misuse returns a status code, never aborts."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_invalid_arguments_return_codes(tmp_path):
    from mitsuba_customization_amd import host, synth
    with host.MerlHip(0) as g:
        L, ctx = g._lib, g._ctx
        assert L.mrl_set_option(ctx, host.OPT_LOOKUP, 7) == -1 and L.mrl_set_option(ctx, 99, 0) == -1
        v = C.c_int()
        assert L.mrl_get_option(ctx, 99, C.byref(v)) == -1 and L.mrl_get_option(ctx, host.OPT_LOOKUP, None) == -1
        mid = C.c_int()
        eta = (C.c_float * 3)(1, 1, 1)
        assert L.mrl_material_ggx(ctx, 0.0, eta, eta, C.byref(mid)) == -1            # alpha must be positive
        assert L.mrl_material_ggx(ctx, 0.1, None, eta, C.byref(mid)) == -1
        tab = np.ones((3, 2, 2, 2))
        sc = (C.c_double * 3)(1, 1, 1)
        assert L.mrl_material_upload_table(ctx, tab.ctypes.data, (C.c_int * 3)(2, 0, 2), sc, C.byref(mid)) == -1
        assert L.mrl_material_upload_table(ctx, None, (C.c_int * 3)(2, 2, 2), sc, C.byref(mid)) == -1
        assert L.mrl_material_upload_f64(ctx, None, C.byref(mid)) == -1
        assert L.mrl_material_load_table(ctx, b"/nonexistent", sc, C.byref(mid)) == -3
        bad = tmp_path / "garbage.binary"
        bad.write_bytes(b"\\x01\\x02\\x03")
        assert L.mrl_material_load_merl(ctx, str(bad).encode(), C.byref(mid)) == -4
        assert L.mrl_material_info(ctx, 0, None, None) == -6                           # no material yet
        z = np.zeros((4, 3), np.float32)
        assert L.mrl_eval_batch(ctx, z.ctypes.data, z.ctypes.data, None, 0, 4, z.ctypes.data) == -6   # no material loaded
        ok = g.upload_merl(synth.make_table("constant"))
        assert L.mrl_eval_batch(ctx, None, z.ctypes.data, None, ok, 4, z.ctypes.data) == -1
        assert L.mrl_eval_batch(ctx, z.ctypes.data, z.ctypes.data, None, ok, 4, None) == -1
        assert L.mrl_eval_batch(ctx, None, None, None, ok, 0, None) == 0                 # empty batch: nothing to check
        assert L.mrl_generate_pairs(ctx, 1, 0, 4, z.ctypes.data, z.ctypes.data, z.ctypes.data) == -1   # generator wants device pointers
        assert L.mrl_generate_materials(ctx, 1, 0, 4, 0, None) == -1
        assert L.mrl_timer_stop(ctx, None) == -1 and L.mrl_device_alloc(ctx, 16, None) == -1
        assert b"alpha" in L.mrl_last_error(ctx) or len(L.mrl_last_error(ctx)) > 0
        # the context still works after all of that
        out = g.eval(np.array([[0, 0, 1]], np.float32), np.array([[0, 0.6, 0.8]], np.float32), material=ok)
        assert out[0, 0] > 0
