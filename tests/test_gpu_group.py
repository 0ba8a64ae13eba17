"""Native multi-device host path (mrl_group_*) on the 1-GPU box.  One member: the sharded call IS the single-device
call.  Several members that name GPU 0 again: the whole pipeline — index tiles, per-member streams, double-buffered
chunks, transfers behind the computes, ordering of the root's stream — runs with device copies as the transport;
results must equal a single-device run over the same unit range bit for bit (they are a pure function of the unit
index).  RCCL itself needs >= 2 GPUs: here only its loading + communicator set-up (one rank) is exercised; the
point-to-point leg is UNMEASURED on N>1 hardware."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mitsuba_customization_amd", "lib")


def _outs(n, dev):
    import torch
    return (torch.full((n, 3), -7.0, device=dev), torch.full((n,), -7.0, device=dev), torch.full((n, 3), -7.0, device=dev),
            torch.full((n,), -7.0, device=dev), torch.full((n, 3), -7.0, device=dev))


def _single_device_reference(tabs, n, n_materials, seed=0x5EED, first=0):
    from mitsuba_customization_amd import host
    with host.MerlHip(0) as g:
        ids = [g.upload_merl(t) for t in tabs]
        wi, wo, u = g.generate_pairs(seed, first, n)
        mat = g.generate_materials(seed, first, n, n_materials) if n_materials else None
        return [t.clone() for t in g.eval_sample(wi, wo, u, mat=mat, material=ids[0])]


def test_one_member_is_the_single_device_call(tables):
    import torch
    from mitsuba_customization_amd import host
    n = 300_001
    ref = _single_device_reference([tables("ggx_tab", 0)], n, 0)
    with host.MerlGroup([0]) as grp:
        assert grp.size == 1 and grp.transport == host.TRANSPORT_PEER_COPY
        mid = grp.upload_merl(tables("ggx_tab", 0))
        tiles = grp.generate_tiles(0x5EED, 0, n)
        for chunk in (n, 70_000, 1):                        # one launch, several chunks; chunk = 1 only on a small prefix
            m = n if chunk > 1 else 37
            out = _outs(m, torch.device("cuda", 0))
            grp.eval_sample_sharded(tiles, m, chunk, out, root=0, material=mid)
            grp.synchronize()
            for a, b in zip(out, ref):
                assert torch.equal(a.view(torch.int32), b[:m].view(torch.int32))
        assert grp.last_timing()[0] > 0.0


@pytest.mark.parametrize("members,root,n,chunk,n_tables", [(2, 0, 250_003, 60_000, 1), (3, 2, 100_001, 9_000, 3), (4, 1, 5, 2, 1), (3, 0, 2, 64, 1)])
def test_members_sharing_the_gpu_reproduce_the_single_device_run(tables, members, root, n, chunk, n_tables):
    import torch
    from mitsuba_customization_amd import host
    tabs = [tables("ggx_tab", s) for s in range(n_tables)]
    ref = _single_device_reference(tabs, n, n_tables if n_tables > 1 else 0)
    with host.MerlGroup([0] * members) as grp:
        assert grp.transport == host.TRANSPORT_PEER_COPY
        ids = [grp.upload_merl(t) for t in tabs]
        assert ids == list(range(n_tables))
        tiles = grp.generate_tiles(0x5EED, 0, n, n_tables if n_tables > 1 else 0)
        out = _outs(n, torch.device("cuda", 0))
        for _ in range(3):                                  # repeated calls reuse the double buffers
            grp.eval_sample_sharded(tiles, n, chunk, out, root=root, material=ids[0])
        grp.synchronize()
        for a, b in zip(out, ref):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        assert len(grp.last_timing()) == members
        # eval only: one result array through the same pipeline (alternating with the fused call on the same buffers)
        rgb = torch.full((n, 3), -7.0, device="cuda")
        grp.eval_sharded(tiles, n, chunk, rgb, root=root, material=ids[0])
        grp.eval_sample_sharded(tiles, n, chunk, out, root=root, material=ids[0])
        grp.eval_sharded(tiles, n, max(1, chunk // 2), rgb, root=root, material=ids[0])
        grp.synchronize()
        with host.MerlHip(0) as single:
            sid = [single.upload_merl(t) for t in tabs]
            wi, wo, u = single.generate_pairs(0x5EED, 0, n)
            mat = single.generate_materials(0x5EED, 0, n, n_tables) if n_tables > 1 else None
            want = single.eval(wi, wo, mat=mat, material=sid[0])
            assert torch.equal(rgb.view(torch.int32), want.view(torch.int32))
        for a, b in zip(out, ref):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_an_rgl_material_is_replicated_and_sharded_like_a_table():
    """The adaptive-parameterisation material through the device-group pipeline (three members sharing GPU 0, root 1, uneven
    chunks): bit-identical to the single-device call."""
    import torch
    from mitsuba_customization_amd import host, synth
    fields = synth.make_rgl_fields(seed=13, n_phi=1, n_theta=5, res=10)
    n = 120_007
    with host.MerlHip(0) as g:
        mid = g.upload_rgl(fields)
        wi, wo, u = g.generate_pairs(0x5EED, 0, n)
        ref = [t.clone() for t in g.eval_sample(wi, wo, u, material=mid)]
    with host.MerlGroup([0, 0, 0]) as grp:
        ggx = grp.ggx(0.2, (1.5, 1.5, 1.5), (3.0, 3.0, 3.0))
        mid = grp.upload_rgl(fields)
        assert (ggx, mid) == (0, 1)
        tiles = grp.generate_tiles(0x5EED, 0, n)
        out = _outs(n, torch.device("cuda", 0))
        grp.eval_sample_sharded(tiles, n, 17_000, out, root=1, material=mid)
        grp.synchronize()
        for a, b in zip(out, ref):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        grp.release_material(mid)
        assert grp.upload_rgl(fields) == mid


def test_host_arrays_split_over_members(oracle, tables):
    from mitsuba_customization_amd import host
    n = 200_003
    wi, wo, u = oracle.generate_pairs(0x5EED, 5, n)
    with host.MerlHip(0) as g:
        mid = g.upload_merl(tables("ggx_tab", 0))
        ref = g.eval_sample(wi, wo, u, material=mid)
    with host.MerlGroup([0, 0, 0]) as grp:
        mid = grp.upload_merl(tables("ggx_tab", 0))
        out = grp.eval_sample_host(wi, wo, u, material=mid)
    for a, b in zip(out, ref):
        assert np.array_equal(a.view(np.int32), b.view(np.int32))
    # the single-purpose calls split the same way
    with host.MerlGroup([0, 0]) as grp:
        mid = grp.upload_merl(tables("ggx_tab", 0))
        ev, pd = grp.eval_host(wi, wo, material=mid), grp.pdf_host(wi, wo, material=mid)
        ep = grp.eval_pdf_host(wi, wo, material=mid)
        sm = grp.sample_host(wi, u, material=mid)
    assert np.array_equal(ev, ref[0]) and np.array_equal(pd, ref[1]) and np.array_equal(ep[0], ref[0]) and np.array_equal(ep[1], ref[1])
    assert np.array_equal(sm[0], ref[2]) and np.array_equal(sm[1], ref[3]) and np.array_equal(sm[2], ref[4])
    want = oracle.eval_sample_multi([oracle.OracleTable(tables("ggx_tab", 0))], wi, wo, u, None)
    assert (np.abs(out[0].astype(np.float64) - want[0]) <= 1e-6 * np.abs(want[0]) + 1e-30).all()


def test_group_errors_and_replicated_materials(tables):
    import torch
    from mitsuba_customization_amd import host
    with host.MerlGroup([0, 0]) as grp:
        L = grp._lib
        a = grp.upload_merl(tables("ggx_tab", 0))
        b = grp.ggx(0.1, (1, 1, 1), (2, 2, 2))
        assert (a, b) == (0, 1)
        grp.release_material(a)
        assert grp.upload_table(tables("noise", 3, (8, 8, 16))) == a          # the freed slot, on every member
        grp.set_option(host.OPT_LOOKUP, 1)
        with pytest.raises(host.MerlHipError):
            grp.set_option(host.OPT_LOOKUP, 9)
        tiles = grp.generate_tiles(1, 0, 1000)
        out = _outs(1000, torch.device("cuda", 0))
        with pytest.raises(host.MerlHipError):
            grp.eval_sample_sharded(tiles, 1000, 0, out)                       # chunk of zero units
        with pytest.raises(host.MerlHipError):
            grp.eval_sample_sharded(tiles, 1000, 100, out, root=5)
        with pytest.raises(host.MerlHipError) as e:
            grp.eval_sample_sharded(tiles, 1000, 100, out, material=99)
        assert e.value.status == host.ERR_MATERIAL and "member 0" in str(e.value)
        grp.eval_sample_sharded(tiles, 0, 100, [o[:0] for o in out])           # nothing to do
        grp.synchronize()


def test_rccl_loads_and_builds_a_communicator():
    """One rank: librccl is found, ncclCommInitAll succeeds; there is nobody to send to, so no point-to-point call
    runs — that leg needs >= 2 GPUs (bench.py runs it through lib/group_host when it is given N > 1 GPUs)."""
    import torch
    from mitsuba_customization_amd import host, synth
    with host.MerlGroup([0], transport=host.TRANSPORT_RCCL) as grp:
        assert grp.transport == host.TRANSPORT_RCCL
        mid = grp.upload_merl(synth.make_table("ggx_tab", 0))
        tiles = grp.generate_tiles(0x5EED, 0, 10_000)
        out = _outs(10_000, torch.device("cuda", 0))
        grp.eval_sample_sharded(tiles, 10_000, 3_000, out, material=mid)
        grp.synchronize()
        assert float(out[0].min()) >= 0.0 and float(out[0].max()) > 0.0


def test_native_cpp_host_program():
    """examples/group_host.cpp: the C++ host over mrl_group_* — three members on GPU 0, 16 mixed tables, gather by
    device copies, checked inside the program against a single-device run."""
    r = subprocess.run([os.path.join(LIB, "group_host"), "--devices", "0,0,0", "--units-per-device", str(1 << 20), "--chunk", str(300_000),
                        "--tables", "16", "--steps", "2", "--warmup", "1", "--check", "--root", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["check_mismatches"] == 0 and out["transport"] == "peer_copy" and out["devices"] == [0, 0, 0]
    assert out["compute_only_Meval_s"] > 0 and out["gathered_Meval_s"] > 0
    # the same program, no Python and no torch in the process: librccl is found by the library's own dlopen and a
    # one-rank communicator is built (what the N-GPU run of bench.py's "native_group" leg starts with)
    r = subprocess.run([os.path.join(LIB, "group_host"), "--devices", "0", "--transport", "rccl", "--units-per-device", str(1 << 20),
                        "--steps", "1", "--warmup", "1", "--check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["transport"] == "rccl" and out["check_mismatches"] == 0


def test_link_test_and_rgb_only_gather_on_the_rehearsal_transport():
    """mrl_group_link_test: every peer -> root link on its own, timed and bit-checked (device copies here: members share
    GPU 0; the RCCL branch needs distinct devices and is refused on this group)."""
    from mitsuba_customization_amd import host
    with host.MerlGroup([0, 0, 0]) as grp:
        assert grp.transport == host.TRANSPORT_PEER_COPY
        for nbytes in (1 << 20, 16 << 20):
            rep = grp.link_test(nbytes, root=1)
            assert [r["peer"] for r in rep] == [0, 1, 2]
            assert all(r["ok"] and r["mismatches"] == 0 for r in rep)
            assert rep[1]["GBps"] == 0.0 and rep[0]["GBps"] > 1.0 and rep[2]["GBps"] > 1.0
        with pytest.raises(host.MerlHipError) as e:
            grp.link_test(1 << 20, transport=host.TRANSPORT_RCCL)
        assert e.value.status == host.ERR_COMM
        with pytest.raises(host.MerlHipError):
            grp.link_test(2)


def test_native_host_selftest_and_fallback_to_device_copies_in_a_fresh_process():
    """group_host --selftest --transport rccl over members that share GPU 0: RCCL cannot serve a repeated device, the child
    running that leg fails, and the parent repeats the run with device copies in a FRESH child — the branch a real
    multi-GPU run takes when RCCL fails.  The line says what failed; three rates are reported side by side."""
    exe = os.path.join(LIB, "group_host")
    common = ["--devices", "0,0", "--units-per-device", str(1 << 20), "--chunk", str(400_000), "--steps", "2", "--warmup", "1", "--check", "--selftest"]
    r = subprocess.run([exe] + common + ["--transport", "rccl"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "repeating it with device copies in a fresh process" in r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["transport"] == "peer_copy" and out["fallback_from"].startswith("rccl exit code") and out["check_mismatches"] == 0
    assert out["compute_only_Meval_s"] > 0 and out["rgb_gathered_Meval_s"] > 0 and out["gathered_Meval_s"] > 0
    links = out["selftest"]["peer_copy"]
    assert [row["bytes"] for row in links] == [1 << 20, 4 << 20, 16 << 20, 64 << 20]
    assert all(len(row["GBps_per_peer"]) == 1 and row["GBps_per_peer"][0] > 1.0 and row["mismatches"] == 0 for row in links)
    # without the fallback the failure is the exit code
    r = subprocess.run([exe] + common + ["--transport", "rccl", "--no-fallback"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "members" not in r.stdout
    # AUTO picks device copies for a repeated device by itself: one child, no fallback
    r = subprocess.run([exe] + common, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["fallback_from"] is None and out["transport"] == "peer_copy"
