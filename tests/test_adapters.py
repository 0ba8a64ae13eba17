"""Plugin adapters (Mitsuba 0.6 / Mitsuba 3 class interfaces over the C ABI).

CPU part: the plugin shared objects build, load and export the entry points the hosts' plugin
managers resolve; without a GPU the constructor fails loudly (no CPU fallback).
GPU part: a driver that plays plugin manager + integrator calls the scalar virtual interface and
the batched (wavefront) interface; both are compared with the oracle by this test.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from mitsuba_customization_amd import build, synth

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mitsuba_customization_amd")
LIB = os.path.join(PKG, "lib")
# The plugins' `scalar` property: "cpu" (default) evaluates the virtual per-ray calls on the calling thread, "gpu" sends them
# through the device's one-unit call service.  Only the latter runs the batch kernels' code on the batch kernels' hardware
# and can be held to bit-for-bit agreement with the batch calls; the CPU path has its own test below.
GPU_SCALAR = dict(os.environ, MERL_DRIVER_SCALAR="gpu")


@pytest.fixture(scope="module")
def built():
    build.build_all()
    return LIB


@pytest.fixture(scope="module")
def merl_file(tmp_path_factory, tables):
    p = str(tmp_path_factory.mktemp("merl") / "synthetic_ggx_tab.binary")
    synth.write_merl_binary(p, tables("ggx_tab", 0))
    return p


def test_plugins_export_host_entry_points(built):
    for name in ("merl", "customized_measurement"):
        so06 = C.CDLL(os.path.join(built, "plugins06", name + ".so"))
        assert hasattr(so06, "CreateInstance") and hasattr(so06, "GetDescription")
        so06.GetDescription.restype = C.c_char_p
        assert b"libmerl_hip" in so06.GetDescription()
        so3 = C.CDLL(os.path.join(built, "plugins3", name + ".so"))
        for sym in ("plugin_name", "plugin_descr", "plugin_create_scalar_rgb"):
            assert hasattr(so3, sym)
        so3.plugin_name.restype = C.c_char_p
        assert so3.plugin_name() in (b"MerlBSDF", b"CustomizedMeasurement")
    so3 = C.CDLL(os.path.join(built, "plugins3", "measured.so"))
    so3.plugin_name.restype = C.c_char_p
    assert so3.plugin_name() == b"Measured" and hasattr(so3, "plugin_create_scalar_rgb")


def test_plugin_constructor_fails_loudly_without_gpu(built, merl_file):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    for drv, plug in (("driver06", "plugins06/merl.so"), ("driver3", "plugins3/merl.so")):
        r = subprocess.run([os.path.join(built, drv), "--expect-no-device", os.path.join(built, plug), merl_file],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "no CPU fallback" in r.stdout


def _write_pairs(path, wi, wo, u):
    with open(path, "wb") as f:
        np.asarray([wi.shape[0]], np.uint64).tofile(f)
        wi.tofile(f); wo.tofile(f); u.tofile(f)


def _read_out(path, m, n):
    a = np.fromfile(path, np.float32)
    assert a.size == 11 * (m + n)
    return a[:11 * m].reshape(m, 11), a[11 * m:].reshape(n, 11)


def _check(out, want, exact_dirs=True):
    rgb, pdf, wo2, pdf2, w = want
    tol = lambda g, r: (np.abs(g.astype(np.float64) - r) <= 1e-6 * np.abs(r) + 1e-30).all()
    assert tol(out[:, 0:3], rgb), "rgb"
    assert np.array_equal(out[:, 3], pdf) and np.array_equal(out[:, 4:7], wo2) and np.array_equal(out[:, 7], pdf2)
    assert tol(out[:, 8:11], w), "weight"


@pytest.mark.gpu
@pytest.mark.parametrize("host,disk", [("06", 0), ("3", 1)])
@pytest.mark.parametrize("interp", ["trilinear", "nearest"])
def test_merl_plugin_scalar_and_batched_calls_match_oracle(built, merl_file, oracle, tables, tmp_path, host, disk, interp):
    n, m = 20000, 300
    wi, wo, u = oracle.generate_pairs(0x5EED, 424242, n)
    wi[3, 2] = -wi[3, 2]; wo[5, 2] = -wo[5, 2]                       # below-horizon guards
    pairs, out = str(tmp_path / "pairs.bin"), str(tmp_path / "out.bin")
    _write_pairs(pairs, wi, wo, u)
    drv = os.path.join(built, "driver06" if host == "06" else "driver3")
    plug = os.path.join(built, "plugins06" if host == "06" else "plugins3", "merl.so")
    r = subprocess.run([drv, plug, merl_file, pairs, out, str(m), interp], capture_output=True, text=True, timeout=300, env=GPU_SCALAR)
    assert r.returncode == 0, r.stdout + r.stderr
    scalar, batch = _read_out(out, m, n)
    assert np.array_equal(scalar, batch[:m]), "scalar virtual calls and the batch path must agree bit for bit"
    lookup = 1 if interp == "trilinear" else 0
    want = oracle.eval_sample_multi([oracle.OracleTable(tables("ggx_tab", 0))], wi, wo, u, None,
                                    oracle.make_opts(lookup=lookup, disk_map=disk))
    if lookup:
        _check(batch, want)
    else:
        ok = np.abs(batch[:, 0:3].astype(np.float64) - want[0]) <= 1e-6 * np.abs(want[0]) + 1e-30
        assert (~ok.all(axis=1)).sum() <= 1                     # nearest: at most one bin-edge flip (measured: none in 67 M lookups)


@pytest.mark.gpu
@pytest.mark.parametrize("host,disk", [("06", 0), ("3", 1)])
def test_customized_measurement_plugin(built, oracle, tables, tmp_path, host, disk):
    dims, scale = (24, 40, 60), (0.5, 2.0, 1.25)
    tab = tables("noise", 77, dims)
    tfile = str(tmp_path / "custom.binary")
    synth.write_merl_binary(tfile, tab)
    n, m = 8000, 100
    wi, wo, u = oracle.generate_pairs(0x5EED, 99, n)
    pairs, out = str(tmp_path / "pairs.bin"), str(tmp_path / "out.bin")
    _write_pairs(pairs, wi, wo, u)
    drv = os.path.join(built, "driver06" if host == "06" else "driver3")
    plug = os.path.join(built, "plugins06" if host == "06" else "plugins3", "customized_measurement.so")
    r = subprocess.run([drv, plug, tfile, pairs, out, str(m), "trilinear"] + [str(s) for s in scale],
                       capture_output=True, text=True, timeout=300, env=GPU_SCALAR)
    assert r.returncode == 0, r.stdout + r.stderr
    scalar, batch = _read_out(out, m, n)
    assert np.array_equal(scalar, batch[:m])
    want = oracle.eval_sample_multi([oracle.OracleTable(tab, scale)], wi, wo, u, None, oracle.make_opts(disk_map=disk))
    _check(batch, want)


@pytest.mark.gpu
@pytest.mark.parametrize("host,disk", [("06", 0), ("3", 1)])
@pytest.mark.parametrize("name,param", [("standard", 1), ("standard_full", 2)])
def test_customized_measurement_parameterization_property(built, oracle, tables, tmp_path, host, disk, name, param):
    """<string name="parameterization" value="standard"/>: the table is indexed by (theta_i, theta_o, dphi) — enum mrl_param.
    The 0.6 driver also serialises the instance and evaluates the unserialised copy (the property travels with it)."""
    dims, scale = (20, 16, 36), (0.5 / 1024, 2.0 / 1024, 1.25 / 1024)          # exact in Float: 0.6 keeps its scales as Float
    tab = tables("ggx_std" if param == 1 else "ggx_std_full", 6, dims)
    tfile = str(tmp_path / "custom.binary")
    synth.write_merl_binary(tfile, tab)
    n, m = 6000, 60
    wi, wo, u = oracle.generate_pairs(0x5EED, 123, n)
    pairs, out = str(tmp_path / "pairs.bin"), str(tmp_path / "out.bin")
    _write_pairs(pairs, wi, wo, u)
    drv = os.path.join(built, "driver06" if host == "06" else "driver3")
    plug = os.path.join(built, "plugins06" if host == "06" else "plugins3", "customized_measurement.so")
    r = subprocess.run([drv, plug, tfile, pairs, out, str(m), "trilinear"] + [repr(float(s)) for s in scale] + ["cosine", name],
                       capture_output=True, text=True, timeout=300, env=GPU_SCALAR)
    assert r.returncode == 0, r.stdout + r.stderr
    scalar, batch = _read_out(out, m, n)
    assert np.array_equal(scalar, batch[:m])
    want = oracle.eval_sample_multi([oracle.OracleTable(tab, scale, param=param)], wi, wo, u, None, oracle.make_opts(disk_map=disk))
    _check(batch, want)
    # the same file under the default parameterisation is a different material (and a different resident table)
    r = subprocess.run([drv, plug, tfile, pairs, out, str(m), "trilinear"] + [repr(float(s)) for s in scale], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    _, other = _read_out(out, m, n)
    assert not np.array_equal(other[0], batch[0])
    r = subprocess.run([drv, plug, tfile, pairs, out, str(m), "trilinear"] + [repr(float(s)) for s in scale] + ["cosine", "polar"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "parameterization" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("host", ["06", "3"])
def test_customized_measurement_reads_a_tensor_file_table(built, oracle, tables, tmp_path, host):
    """filename="*.bsdf": the table comes out of a tensor_file container (field "table" [3, h, d, p] + "scale")."""
    dims, scale = (16, 12, 20), (0.5, 2.0, 1.25)
    tab = tables("ggx_tab", 4, dims).astype(np.float32)
    tfile = str(tmp_path / "custom.bsdf")
    synth.write_tensor_file(tfile, {"table": tab, "scale": np.array(scale)})
    n, m = 4000, 50
    wi, wo, u = oracle.generate_pairs(0x5EED, 77, n)
    pairs, out = str(tmp_path / "pairs.bin"), str(tmp_path / "out.bin")
    _write_pairs(pairs, wi, wo, u)
    drv = os.path.join(built, "driver06" if host == "06" else "driver3")
    plug = os.path.join(built, "plugins06" if host == "06" else "plugins3", "customized_measurement.so")
    r = subprocess.run([drv, plug, tfile, pairs, out, str(m)], capture_output=True, text=True, timeout=300, env=GPU_SCALAR)
    assert r.returncode == 0, r.stdout + r.stderr
    scalar, batch = _read_out(out, m, n)
    assert np.array_equal(scalar, batch[:m])
    want = oracle.eval_sample_multi([oracle.OracleTable(tab, scale)], wi, wo, u, None, oracle.make_opts(disk_map=0 if host == "06" else 1))
    _check(batch, want)
    # a 5-channel table cannot go through an RGB Spectrum: the constructor says so
    wide = os.path.join(os.path.dirname(__file__), "golden", "tensor_table_c5.bsdf")
    r = subprocess.run([drv, plug, wide, pairs, out, str(m)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 5 and "5 channels" in r.stderr


@pytest.mark.gpu
def test_plugin_reports_missing_file(built, tmp_path):
    r = subprocess.run([os.path.join(built, "driver06"), os.path.join(built, "plugins06", "merl.so"), "/nonexistent.binary",
                        "/dev/null", str(tmp_path / "o"), "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 5 and "cannot open" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("host,disk", [("06", 0), ("3", 1)])
def test_plugin_table_sampling_property(built, merl_file, oracle, tables, tmp_path, host, disk):
    """<string name="sampling" value="table"/> switches sample()/pdf() to table importance sampling."""
    n, m = 6000, 100
    wi, wo, u = oracle.generate_pairs(0x5EED, 2024, n)
    pairs, out = str(tmp_path / "pairs.bin"), str(tmp_path / "out.bin")
    _write_pairs(pairs, wi, wo, u)
    drv = os.path.join(built, "driver06" if host == "06" else "driver3")
    plug = os.path.join(built, "plugins06" if host == "06" else "plugins3", "merl.so")
    r = subprocess.run([drv, plug, merl_file, pairs, out, str(m), "trilinear", "1", "1", "1", "table"], capture_output=True, text=True, timeout=300, env=GPU_SCALAR)
    assert r.returncode == 0, r.stdout + r.stderr
    scalar, batch = _read_out(out, m, n)
    assert np.array_equal(scalar, batch[:m])
    T = oracle.OracleTable(tables("ggx_tab", 0))
    c_wo, c_pdf, c_w = T.sample_table(wi, u, oracle.make_opts(disk_map=disk))
    assert np.abs(batch[:, 4:7].astype(np.float64) - c_wo).max() <= 1.2e-7
    ok = np.abs(batch[:, 7].astype(np.float64) - c_pdf) <= 2e-6 * np.abs(c_pdf) + 1e-30
    assert ok.mean() > 0.999
    assert (np.abs(batch[:, 3].astype(np.float64) - T.pdf_table(wi, wo)) <= 2e-6 * T.pdf_table(wi, wo) + 1e-30).all()


@pytest.mark.gpu
@pytest.mark.parametrize("host,disk", [("06", 0), ("3", 1)])
@pytest.mark.parametrize("props,cosine,negative", [("cosine_factor=omitted", 1, 0), ("negative_values=keep", 0, 1), ("negative_values=renormalize,cosine_factor=omitted", 1, 2)])
def test_plugin_convention_properties(built, merl_file, oracle, tables, tmp_path, host, disk, props, cosine, negative):
    """SURVEY.md Appendix B 4 and 2 as plugin properties: <string name="cosine_factor" value="included|omitted"/> and
    <string name="negative_values" value="clamp|keep|renormalize"/> — scalar virtual calls (on the CPU) and the batch path answer
    with the oracle's values under the same convention."""
    n, m = 8000, 400
    wi, wo, u = oracle.generate_pairs(0x5EED, 777 + cosine + 2 * negative, n)
    pairs, out = str(tmp_path / "pairs.bin"), str(tmp_path / "out.bin")
    _write_pairs(pairs, wi, wo, u)
    drv = os.path.join(built, "driver06" if host == "06" else "driver3")
    plug = os.path.join(built, "plugins06" if host == "06" else "plugins3", "merl.so")
    r = subprocess.run([drv, plug, merl_file, pairs, out, str(m), "trilinear"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, MERL_DRIVER_PROPS=props))
    assert r.returncode == 0, r.stdout + r.stderr
    scalar, batch = _read_out(out, m, n)
    want = oracle.eval_sample_multi([oracle.OracleTable(tables("ggx_tab", 0))], wi, wo, u, None,
                                    oracle.make_opts(lookup=1, disk_map=disk, cosine=cosine, negative=negative))
    default = oracle.eval_sample_multi([oracle.OracleTable(tables("ggx_tab", 0))], wi, wo, u, None, oracle.make_opts(lookup=1, disk_map=disk))
    assert not np.array_equal(want[0], default[0])                  # the property changes the answer
    _check(batch, want)
    _check(scalar, tuple(w[:m] for w in want))
    r = subprocess.run([drv, plug, merl_file, pairs, out, str(m), "trilinear"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, MERL_DRIVER_PROPS="negative_values=sometimes"))
    assert r.returncode != 0 and "negative_values" in (r.stdout + r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("host", ["06", "3"])
def test_plugin_instances_share_one_resident_table_and_release_it(built, merl_file, tables, tmp_path, host):
    """Two <bsdf> elements naming the same .binary hold ONE table in HBM; 50 create/destroy cycles leave free memory flat."""
    other = str(tmp_path / "other.binary")
    synth.write_merl_binary(other, tables("ggx_tab", 1))
    drv = os.path.join(built, "driver06" if host == "06" else "driver3")
    plug = os.path.join(built, "plugins06" if host == "06" else "plugins3", "merl.so")
    r = subprocess.run([drv, plug, merl_file, "--residency", other, "50"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "residency ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("host", ["06", "3"])
def test_plugin_resolves_scene_relative_filename(built, merl_file, oracle, tmp_path, host):
    """filename="synthetic_ggx_tab.binary" + the scene's directory on the host's FileResolver search path."""
    n, m = 2000, 20
    wi, wo, u = oracle.generate_pairs(0x5EED, 1, n)
    pairs, out_rel, out_abs = str(tmp_path / "pairs.bin"), str(tmp_path / "rel.bin"), str(tmp_path / "abs.bin")
    _write_pairs(pairs, wi, wo, u)
    drv = os.path.join(built, "driver06" if host == "06" else "driver3")
    plug = os.path.join(built, "plugins06" if host == "06" else "plugins3", "merl.so")
    env = dict(os.environ, MITSUBA_MIRROR_DATA_PATH="/nonexistent-dir:" + os.path.dirname(merl_file))
    r = subprocess.run([drv, plug, os.path.basename(merl_file), pairs, out_rel, str(m)], capture_output=True, text=True, timeout=300,
                       env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([drv, plug, merl_file, pairs, out_abs, str(m)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert open(out_rel, "rb").read() == open(out_abs, "rb").read()
    # without the search path the relative name does not resolve: the constructor reports the I/O error
    r = subprocess.run([drv, plug, os.path.basename(merl_file), pairs, out_rel, str(m)], capture_output=True, text=True, timeout=120,
                       cwd=str(tmp_path))
    assert r.returncode == 5 and "cannot open" in r.stderr


@pytest.mark.gpu
def test_c99_example_runs(built, merl_file):
    """examples/abi_example.c: the ABI from plain C — pinned zero-copy vs staged host path, error codes."""
    r = subprocess.run([os.path.join(built, "abi_example"), merl_file], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "example ok" in r.stdout and "agree bit for bit" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_scalar_calls_from_many_threads_are_combined(built, merl_file, oracle, tmp_path):
    """16 render threads calling the scalar virtual interface share GPU rounds instead of queueing on a mutex."""
    import re
    n = m = 16000
    wi, wo, u = oracle.generate_pairs(7, 0, n)
    pairs, out = str(tmp_path / "pairs.bin"), str(tmp_path / "out.bin")
    _write_pairs(pairs, wi, wo, u)
    r = subprocess.run([os.path.join(built, "driver06"), os.path.join(built, "plugins06", "merl.so"), merl_file, pairs, out, str(m)],
                       capture_output=True, text=True, timeout=600, env=GPU_SCALAR)
    assert r.returncode == 0, r.stdout + r.stderr              # includes the bit-for-bit check of every threaded call
    found = re.search(r"scalar calls: ([0-9.e+-]+) us/call from one thread, ([0-9.e+-]+) us/call amortised over 16 threads", r.stdout)
    assert found, r.stdout
    single, combined = float(found.group(1)), float(found.group(2))
    print(f"scalar plugin call: {single:.1f} us single-threaded, {combined:.2f} us amortised over 16 threads")
    assert combined < 0.5 * single


@pytest.mark.gpu
@pytest.mark.parametrize("host,disk,m", [("06", 0, 1 << 20), ("3", 1, 1 << 18)])
@pytest.mark.parametrize("interp", ["trilinear", "nearest"])
def test_scalar_calls_on_the_cpu_match_oracle(built, merl_file, oracle, tables, tmp_path, host, disk, m, interp):
    """BASELINE configs[0] as a product path: 2^20 scalar eval() / pdf() / sample() through the plugin's virtual interface
    with scalar="cpu" (the default) — the kernels' own per-unit functions compiled for the host over a host image of the
    resident table (mrl_host_*, never oracle/).  Against the oracle: values to 1e-6, sampled directions and pdfs bit-identical;
    against the GPU batch call on the same units: one Float ulp at most, and nearly every value the same bits (the two builds
    differ in the hardware reciprocal seeds only).  Parity unpinned: the oracle is this repo's restatement."""
    import re
    n = m
    wi, wo, u = oracle.generate_pairs(0x5EED, 777, n)
    wi[3, 2] = -wi[3, 2]; wo[5, 2] = -wo[5, 2]                       # below-horizon guards
    pairs, out = str(tmp_path / "pairs.bin"), str(tmp_path / "out.bin")
    _write_pairs(pairs, wi, wo, u)
    drv = os.path.join(built, "driver06" if host == "06" else "driver3")
    plug = os.path.join(built, "plugins06" if host == "06" else "plugins3", "merl.so")
    env = dict(os.environ, MERL_DRIVER_SCALAR="cpu")
    r = subprocess.run([drv, plug, merl_file, pairs, out, str(m), interp], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "scalar=cpu" in r.stdout.replace(" ", "")
    scalar, batch = _read_out(out, m, n)
    lookup = 1 if interp == "trilinear" else 0
    want = oracle.eval_sample_multi([oracle.OracleTable(tables("ggx_tab", 0))], wi, wo, u, None,
                                    oracle.make_opts(lookup=lookup, disk_map=disk))
    if lookup:
        _check(scalar, want)
        rel = np.abs(scalar.astype(np.float64) - batch) / np.maximum(np.abs(batch.astype(np.float64)), 1e-30)
        assert rel.max() <= 1.3e-7, "CPU scalar call vs GPU batch call: more than one Float ulp apart"
        same = (scalar == batch).all(axis=1).mean()
        print(f"CPU scalar calls bit-identical to the batch call on {same * 100:.4f} % of {m} units")
        assert same > 0.999
    else:
        ok = np.abs(scalar[:, 0:3].astype(np.float64) - want[0]) <= 1e-6 * np.abs(want[0]) + 1e-30
        assert (~ok.all(axis=1)).sum() <= 1                     # nearest: at most one bin-edge flip
        assert np.array_equal(scalar[:, 3], want[1]) and np.array_equal(scalar[:, 4:7], want[2]) and np.array_equal(scalar[:, 7], want[3])
    if host == "06":
        found = re.search(r"scalar calls: ([0-9.e+-]+) us/call from one thread, ([0-9.e+-]+) us/call amortised over 16 threads", r.stdout)
        assert found, r.stdout
        single = float(found.group(1))
        print(f"CPU scalar plugin call: {single:.3f} us per virtual call from one thread, {float(found.group(2)):.3f} us amortised over 16")
        assert single < 0.5, "a scalar virtual call on the CPU path should cost what the CPU plugin it replaces costs"


@pytest.mark.gpu
@pytest.mark.parametrize("n_phi", [1, 5])
def test_mitsuba3_measured_plugin_evaluates_an_rgl_file(built, tmp_path, n_phi):
    """<bsdf type="measured"> over a synthetic file with the RGL field names (PARITY UNPINNED: no database file and no upstream
    source offline; the checker is oracle/rgl_oracle.c).  Scalar virtual calls run on the CPU (the per-unit functions compiled
    for the host), BatchedBSDF calls on the GPU; both against the oracle at 1e-6, and against each other."""
    from oracle.binding import OracleRgl, generate_pairs
    fields = synth.make_rgl_fields(seed=11, n_phi=n_phi, n_theta=5, res=10)
    tfile = str(tmp_path / "synthetic_rgb.bsdf")
    synth.write_tensor_file(tfile, fields)
    n, m = 6000, 400
    wi, wo, u = generate_pairs(0x5EED, 4711, n)
    wi[3, 2] = -wi[3, 2]; wo[5, 2] = -wo[5, 2]
    pairs, out = str(tmp_path / "pairs.bin"), str(tmp_path / "out.bin")
    _write_pairs(pairs, wi, wo, u)
    drv, plug = os.path.join(built, "driver3"), os.path.join(built, "plugins3", "measured.so")
    r = subprocess.run([drv, plug, tfile, pairs, out, str(m)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "plugin: Measured" in r.stdout and "scalar = cpu" in r.stdout
    scalar, batch = _read_out(out, m, n)
    orc = OracleRgl(fields)
    rgb, pdf = orc.eval_pdf(wi, wo)

    def close(a, b, what):
        b = np.asarray(b, np.float64)
        err = np.abs(a.astype(np.float64) - b) / (np.abs(b) + 0.1 * max(float(np.abs(b).max()), 1e-30))
        assert float(err.max()) < 1e-6, (what, float(err.max()))

    for got, name in ((batch, "batch"), (scalar, "scalar")):
        k = got.shape[0]
        close(got[:, 0:3], rgb[:k], name + " eval"); close(got[:, 3], pdf[:k], name + " pdf")
        live = got[:, 7] > 0
        assert live.mean() > 0.5
        c_rgb, c_pdf = orc.eval_pdf(wi[:k][live], got[live, 4:7])            # the oracle AT the direction the plugin returned
        close(got[live, 7], c_pdf, name + " sample pdf"); close(got[live, 8:11], c_rgb / c_pdf[:, None], name + " weight")
    assert float(np.abs(batch[3]).max()) == 0.0 and float(np.abs(batch[5, 0:4]).max()) == 0.0     # below the horizon
    # the CPU one-unit path and the GPU batch path: the same functions on two targets
    close(scalar[:, 0:4], batch[:m, 0:4], "scalar vs batch")
    assert float(np.abs(scalar[:, 4:7] - batch[:m, 4:7]).max()) < 5e-7
    same = np.mean(scalar.view(np.int32) == batch[:m].view(np.int32))
    assert same > 0.95, same
    # scalar = "gpu" is refused by name; a table container is not an RGL file
    r = subprocess.run([drv, plug, tfile, pairs, out, str(m)], capture_output=True, text=True, timeout=120, env=GPU_SCALAR)
    assert r.returncode == 5 and "scalar" in r.stderr
    table = os.path.join(os.path.dirname(__file__), "golden", "tensor_table_c5.bsdf")
    r = subprocess.run([drv, plug, table, pairs, out, str(m)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 5 and "phi_i" in r.stderr


@pytest.mark.gpu
def test_mitsuba3_measured_plugin_spectral_variant(built, tmp_path):
    """<bsdf type="measured"> over a spectral RGL file in the scalar_spectral variant: every ray carries four wavelengths
    (SurfaceInteraction::wavelengths), eval / sample answer with four values interpolated at them.  Scalar virtual calls (on the
    calling thread over the host image) and the BatchedBSDF call against the oracle.  PARITY UNPINNED (synthetic file)."""
    from mitsuba_customization_amd import synth
    from oracle import binding as ob
    fields = synth.make_rgl_fields(seed=61, n_phi=1, n_theta=5, res=10, res_ndf=12, res_sigma=8, n_wavelengths=13)
    path = str(tmp_path / "synthetic_spec.bsdf")
    synth.write_tensor_file(path, fields)
    n, m = 6000, 500
    wi, wo, u = ob.generate_pairs(0x5EED, 61, n)
    wl = np.random.default_rng(61).uniform(340.0, 1020.0, (n, 4)).astype(np.float32)
    pairs, out = str(tmp_path / "pairs.bin"), str(tmp_path / "out.bin")
    with open(pairs, "wb") as f:
        np.asarray([n], np.uint64).tofile(f); wi.tofile(f); wo.tofile(f); u.tofile(f); wl.tofile(f)
    plug = os.path.join(built, "plugins3", "measured.so")
    r = subprocess.run([os.path.join(built, "driver3_spectral"), plug, path, pairs, out, str(m)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    a = np.fromfile(out, np.float32)
    scalar, batch = a[:13 * m].reshape(m, 13), a[13 * m:].reshape(n, 13)
    B = ob.OracleRgl(fields)
    o_val, o_pdf = B.eval_pdf_spectral(wi, wo, wl)
    close = lambda g, r: bool((np.abs(g.astype(np.float64) - r) <= 1e-6 * np.abs(r) + 1e-30).all())
    for got in (batch, scalar):
        k = got.shape[0]
        assert close(got[:, 0:4], o_val[:k]) and close(got[:, 4], o_pdf[:k])
        live = got[:, 8] > 0
        c_val, c_pdf = B.eval_pdf_spectral(wi[:k][live], got[live, 5:8], wl[:k][live])
        assert close(got[live, 8], c_pdf) and close(got[live, 9:13], c_val / c_pdf[:, None])
    assert np.mean(scalar.view(np.int32) == batch[:m].view(np.int32)) > 0.95        # host build vs device: the reciprocal seeds differ
    # the RGB variant refuses the spectral file, the spectral variant an RGB file
    rgb_path = str(tmp_path / "synthetic_rgb.bsdf")
    synth.write_tensor_file(rgb_path, synth.make_rgl_fields(seed=62, n_phi=1, n_theta=4, res=8))
    r = subprocess.run([os.path.join(built, "driver3_spectral"), plug, rgb_path, pairs, out, "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 5 and "spectral variant needs a spectral file" in r.stderr
    r = subprocess.run([os.path.join(built, "driver3"), plug, path, pairs, out, "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 5 and "RGB variant needs RGB data" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("host", ["06", "3"])
def test_plugins_load_through_the_image_cache(built, merl_file, oracle, tmp_path, host):
    """MERL_IMAGE_CACHE_DIR: the first instance writes the table's device image, later processes become resident from it (no parse, no
    re-layout, no sampling-table kernels) and answer with the same bits; a damaged image is refused by the library and rewritten."""
    n, m = 5000, 50
    wi, wo, u = oracle.generate_pairs(0x5EED, 31337, n)
    pairs = str(tmp_path / "pairs.bin")
    _write_pairs(pairs, wi, wo, u)
    drv = os.path.join(built, "driver06" if host == "06" else "driver3")
    plug = os.path.join(built, "plugins06" if host == "06" else "plugins3", "merl.so")
    cache = tmp_path / "cache"
    cache.mkdir()
    env = dict(os.environ, MERL_IMAGE_CACHE_DIR=str(cache))

    def run(tag, e):
        out = str(tmp_path / (tag + ".bin"))
        r = subprocess.run([drv, plug, merl_file, pairs, out, str(m)], capture_output=True, text=True, timeout=300, env=e)
        assert r.returncode == 0, r.stdout + r.stderr
        return open(out, "rb").read()

    plain = run("plain", dict(os.environ))
    assert not list(cache.iterdir())
    first = run("first", env)
    images = list(cache.iterdir())
    assert len(images) == 1 and images[0].suffix == ".mrlimg" and 24_000_000 < images[0].stat().st_size < 26_000_000
    stamp = images[0].stat().st_mtime_ns
    second = run("second", env)
    assert first == plain and second == plain and images[0].stat().st_mtime_ns == stamp          # read, not rewritten
    data = bytearray(images[0].read_bytes()); data[-100] ^= 1
    images[0].write_bytes(bytes(data))
    third = run("third", env)
    assert third == plain and images[0].read_bytes() != bytes(data)                                # refused (checksum), loaded from the source, rewritten
    os.utime(merl_file)                                                                            # a touched source is another image
    run("fourth", env)
    assert len(list(cache.iterdir())) == 2
