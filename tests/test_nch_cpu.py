"""n-channel tables on the CPU: known-answer tests of the oracle's n-channel restatement (the checker of the GPU
tests), and the tensor_file container reader of the library (host code, no GPU needed).
PARITY UNPINNED: the reference's customized_measurement format is unknown; no RGL file exists offline."""
import os
import struct

import numpy as np
import pytest

from mitsuba_customization_amd import host, synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_three_channel_nch_oracle_is_the_rgb_oracle(oracle):
    tab = synth.make_table("noise", 5, (12, 10, 16))
    scale = (0.5, 2.0, 1.25)
    wi, wo, u = oracle.generate_pairs(0x5EED, 0, 4000)
    for lookup, node, disk in ((1, 0, 0), (1, 1, 1), (0, 0, 0)):
        o = oracle.make_opts(lookup, node, disk)
        rgb = oracle.eval_sample_multi([oracle.OracleTable(tab, scale)], wi, wo, u, None, o)
        nch = oracle.eval_sample_nch([oracle.OracleTableNch(tab, scale)], wi, wo, u, None, o)
        assert all(np.array_equal(a, b) for a, b in zip(rgb, nch))


def test_channels_are_independent_and_scaled(oracle):
    """Channel k of a C-channel table == the 1-channel table made of plane k; a channel scale multiplies its output."""
    dims, C = (10, 8, 12), 7
    tab = synth.make_table_nch("noise", C, 3, dims)
    tab = np.abs(tab)                                           # no negative markers: scaling is then exactly linear per texel
    wi, wo, u = oracle.generate_pairs(0x5EED, 77, 3000)
    full = oracle.eval_sample_nch([oracle.OracleTableNch(tab)], wi, wo, u)
    for k in (0, 3, 6):
        one = oracle.eval_sample_nch([oracle.OracleTableNch(tab[k:k + 1])], wi, wo, u)
        assert np.array_equal(full[0][:, k], one[0][:, 0]) and np.array_equal(full[4][:, k], one[4][:, 0])
        assert np.array_equal(full[2], one[2]) and np.array_equal(full[3], one[3])       # directions / pdf do not depend on the table
    scaled = oracle.eval_sample_nch([oracle.OracleTableNch(tab, [2.0] * C)], wi, wo, u)
    assert np.allclose(scaled[0], 2.0 * full[0], rtol=1e-6)


def test_constant_and_affine_nch_tables(oracle):
    dims, C = (9, 7, 10), 5
    const = np.stack([np.full(dims, 100.0 * (c + 1)) for c in range(C)])
    T = oracle.OracleTableNch(const, [0.01] * C)
    assert np.allclose(T.lookup(0.3, 0.4, 1.1), [1.0 * (c + 1) for c in range(C)], rtol=1e-14)
    wi, wo, u = oracle.generate_pairs(1, 0, 500)
    val = oracle.eval_sample_nch([T], wi, wo, u)[0]
    assert np.allclose(val, wo[:, 2:3] * np.arange(1, C + 1)[None, :], rtol=1e-6)      # f * cos(theta_o)
    aff = synth.make_table_nch("affine", C, 0, dims)
    A = oracle.OracleTableNch(aff)
    xh, xd, xp = 3.25, 2.5, 4.75                                 # interior coordinates: trilinear is exact on an affine table
    th, td, pd = (xh * xh / (dims[0] ** 2)) * np.pi / 2, xd / dims[1] * np.pi / 2, xp / dims[2] * np.pi
    want = [10.0 + c + (1.0 + 0.5 * c) * xh + (0.5 + 0.25 * c) * xd + (0.125 * (c + 1)) * xp for c in range(C)]
    assert np.allclose(A.lookup(th, td, pd), want, rtol=1e-12)


def test_mixed_widths_and_unknown_ids_render_zero(oracle):
    a = oracle.OracleTableNch(np.abs(synth.make_table_nch("noise", 4, 1, (6, 6, 8))))
    b = oracle.OracleTableNch(np.abs(synth.make_table_nch("noise", 2, 2, (6, 6, 8))))
    wi, wo, u = oracle.generate_pairs(5, 0, 300)
    mat = (np.arange(300) % 3).astype(np.int32)                 # 0: 4 channels, 1: 2 channels (wrong width), 2: unknown
    val, pdf, wo2, pdf2, w = oracle.eval_sample_nch([a, b], wi, wo, u, mat, n_ch=4)
    assert (val[mat == 0] > 0).any() and not val[mat != 0].any() and not w[mat != 0].any() and not pdf[mat != 0].any()


def test_nch_sampling_marginal_is_normalised(oracle):
    T = oracle.OracleTableNch(synth.make_table_nch("spectral", 8, 4, (30, 20, 24)))
    sp = T.sampling()
    n = sp.n
    s = np.ctypeslib.as_array(sp.s, (n + 1,)); cdf = np.ctypeslib.as_array(sp.cdf, (n + 1,)); c = np.ctypeslib.as_array(sp.c, (n,))
    assert cdf[0] == 0.0 and cdf[n] == 1.0 and (np.diff(cdf) > 0).all()
    assert abs(np.pi * (c * np.diff(s)).sum() - 1.0) < 1e-12     # integral of p_h over the hemisphere of half vectors


# ---- tensor_file container --------------------------------------------------------------------------------------
def test_tensor_file_roundtrip(tmp_path):
    rng = np.random.default_rng(1)
    fields = {"description": np.frombuffer(b"hello tensor file", dtype=np.uint8),
              "table": rng.random((4, 3, 2, 5)).astype(np.float32), "scale": np.array([1.0, 2.0, 0.5, 4.0]),
              "half": rng.random((3, 3)).astype(np.float16), "ids": np.arange(6, dtype=np.int32).reshape(2, 3),
              "big": np.array([2 ** 40 + 3], dtype=np.uint64), "scalar": np.array(7.5, dtype=np.float64)}
    p = str(tmp_path / "t.bsdf")
    synth.write_tensor_file(p, fields)
    got = host.read_tensor_file(p)
    assert list(got) == list(fields)
    assert bytes(got["description"]) == b"hello tensor file"
    assert np.array_equal(got["table"], fields["table"].astype(np.float64)) and got["table"].shape == (4, 3, 2, 5)
    assert np.array_equal(got["scale"], fields["scale"]) and np.array_equal(got["half"], fields["half"].astype(np.float64))
    assert np.array_equal(got["ids"].view(np.int32), fields["ids"]) and int(got["big"][0]) == 2 ** 40 + 3
    assert got["scalar"].shape == () and float(got["scalar"]) == 7.5


def test_committed_tensor_fixture_reads_back():
    got = host.read_tensor_file(os.path.join(GOLDEN, "tensor_table_c5.bsdf"))
    assert got["table"].shape == (5, 6, 5, 8) and got["scale"].tolist() == [1.0, 0.5, 2.0, 1.5, 0.25]
    assert np.array_equal(got["table"], synth.make_table_nch("spectral", 5, 9, (6, 5, 8)).astype(np.float32).astype(np.float64))
    assert np.allclose(got["wavelengths"], np.linspace(400, 700, 5)) and bytes(got["description"]).startswith(b"5-channel")


def test_tensor_file_rejects_malformed_input(tmp_path):
    import ctypes as C
    L = host.load_library()

    def status(data: bytes):
        p = tmp_path / "bad.bsdf"
        p.write_bytes(data)
        f = C.c_void_p()
        rc = L.mrl_tensor_file_open(str(p).encode(), C.byref(f))
        if rc == 0:
            L.mrl_tensor_file_close(f)
        return rc, L.mrl_tensor_file_last_error(None)

    good = tmp_path / "good.bsdf"
    synth.write_tensor_file(str(good), {"table": np.zeros((1, 2, 2, 2), np.float32)})
    data = good.read_bytes()
    assert status(data)[0] == 0
    assert status(b"")[0] == host.ERR_FORMAT and status(b"tensor_fil")[0] == host.ERR_FORMAT
    assert status(b"TENSOR_FILE\0" + data[12:])[0] == host.ERR_FORMAT                      # bad magic
    assert status(data[:12] + b"\x02\x00" + data[14:])[0] == host.ERR_FORMAT               # version 2.0
    assert status(data[:30])[0] == host.ERR_FORMAT                                         # field table cut off
    assert status(data[:-8])[0] == host.ERR_FORMAT                                         # payload outside the file
    bad_dtype = bytearray(data); bad_dtype[18 + 2 + 5 + 2] = 99                            # the field's dtype byte
    rc, msg = status(bytes(bad_dtype))
    assert rc == host.ERR_FORMAT and b"dtype" in msg
    huge = bytearray(data); huge[18 + 2 + 5 + 2 + 1 + 8:18 + 2 + 5 + 2 + 1 + 16] = struct.pack("<Q", 2 ** 60)   # shape[0]
    assert status(bytes(huge))[0] == host.ERR_FORMAT
    f = C.c_void_p()
    assert L.mrl_tensor_file_open(b"/nonexistent.bsdf", C.byref(f)) == host.ERR_IO
    assert L.mrl_tensor_file_open(None, C.byref(f)) == host.ERR_INVALID and L.mrl_tensor_file_field_count(None) == host.ERR_INVALID
