"""bench.py's own launcher: `python bench.py --gpus N` (no torchrun, no rendezvous in the environment) starts the N
ranks itself, relays rank 0's single JSON line and propagates the ranks' return code."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return env


def test_self_launch_propagates_rank_failure():
    """Without a GPU every rank exits non-zero ("needs a GPU"): the launcher must not report a clean run or print a line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by test_self_launch_two_ranks_share_gpu")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--units", "1024"],
                       capture_output=True, text=True, timeout=600, env=_clean_env())
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr
    assert '"metric"' not in r.stdout


def test_launcher_does_not_touch_torch_before_spawning():
    """The parent of a self-launched run must stay GPU-free (never exec or fork a process that initialised HIP):
    bench.self_launch runs with torch import blocked."""
    code = (
        "import sys, types\n"
        "class Block:\n"
        "    def find_spec(self, name, path=None, target=None):\n"
        "        if name == 'torch' or name.startswith('torch.'):\n"
        "            raise ImportError('torch imported in the launcher process')\n"
        "sys.meta_path.insert(0, Block())\n"
        f"sys.argv = ['bench.py', '--gpus', '2', '--launch-timeout', '600', '--units', '1024', '--steps', '1', '--warmup', '0']\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "try:\n"
        "    bench.main()\n"
        "except SystemExit as e:\n"
        "    print('launcher-exit', e.code)\n"
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=_clean_env())
    assert "launcher-exit" in r.stdout, r.stdout + r.stderr
    assert "torch imported in the launcher process" not in r.stderr


@pytest.mark.gpu
def test_self_launch_two_ranks_share_gpu():
    """Plain `python bench.py --gpus 2` end to end on the 1-GPU box: two ranks share GPU 0 for the compute, gloo is the
    control plane and carries the gather leg; one JSON line, n_gpus 2, twice one rank's units."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--share-gpu", "--units", str(1 << 20),
                        "--steps", "3", "--warmup", "1", "--gather-units", str(1 << 18), "--configs4-units", str(6 << 20)],
                       capture_output=True, text=True, timeout=900, env=_clean_env())
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["units_per_gpu_per_step"] == 1 << 20
    assert out["value"] > 0 and out["roofline"]["bytes_per_unit"] == 76
    assert out["parity"]["max_rel_err_vs_oracle"] <= 1e-6
    assert "ms" in out["gather"] and out["gather"]["bytes_into_root"] == 44 * (1 << 18)
    assert "cpu_baseline" not in out                                 # N=1 only
    ng = out["native_group"]                                          # the C++ host over the same "GPUs" (here: GPU 0 twice)
    assert ng["check_mismatches"] == 0 and ng["devices"] == [0, 0] and ng["transport"] == "peer_copy", ng
    # the full-shape configs[4] block, rehearsed at reduced units: 100 resident tables, both CU reservations, every leg checked
    c4 = out["configs4"]
    assert c4["total_units"] == 6 << 20 and c4["units_per_device"] == 3 << 20 and not c4["all_legs_failed"], c4
    assert [(g["transport_asked"], g["reserved_cus"]) for g in c4["legs"]] == [("copy", 0), ("copy", 8)]
    for g in c4["legs"]:
        assert "failed" not in g, g
        assert g["tables_resident"] == 100 and g["check_mismatches"] == 0 and g["transport"] == "peer_copy"
        assert g["compute_only_Meval_s"] > 0 and g["rgb_gathered_Meval_s"] > 0 and g["gathered_Meval_s"] > 0
        assert "peer_copy" in g["selftest"]


@pytest.mark.gpu
def test_resident100_two_ranks_share_gpu():
    """BASELINE configs[4]'s per-device share through the Python ranks: `--config resident100 --gpus 2 --share-gpu` (100 resident
    tables per rank, material ids in the batch: 80 B/unit), without the native legs."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--share-gpu", "--config", "resident100", "--units", str(1 << 20),
                        "--steps", "2", "--warmup", "1", "--gather-units", str(1 << 18), "--no-native-group"],
                       capture_output=True, text=True, timeout=900, env=_clean_env())
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["config"]["materials_resident"] == 100 and out["roofline"]["bytes_per_unit"] == 80
    assert out["parity"]["max_rel_err_vs_oracle"] <= 1e-6 and "ms" in out["gather"]


@pytest.mark.gpu
def test_single_gpu_line_keeps_its_contract():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--units", str(1 << 20), "--steps", "3", "--warmup", "1", "--cpu-reps", "1"],
                       capture_output=True, text=True, timeout=900, env=_clean_env())
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["vs_baseline"] is None and out["config"]["workload"].startswith("BASELINE configs[1]")
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["single_thread_units"] == 1 << 20 and cb["cpu_model"]
    # a mixed batch carries the 4-byte material id in its algorithmic bytes
    r = subprocess.run([sys.executable, BENCH, "--config", "mixed16_256m", "--units", str(1 << 20), "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=_clean_env())
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["roofline"]["bytes_per_unit"] == 80 and out["config"]["materials_resident"] == 16
    assert out["parity"]["max_rel_err_vs_oracle"] <= 1e-6
