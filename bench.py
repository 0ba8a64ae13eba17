#!/usr/bin/env python3
"""bench.py — BSDF eval+sample throughput on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 --steps 20 --warmup 3          # spawns its own 8 ranks (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (mrl_eval_sample_batch: eval rgb + pdf + sample wo'/pdf'/weight')
over one batch of synthetic (wi, wo, u) that is already resident in HBM.  The workload is
BASELINE.json configs[1]: single MERL material, 64M pairs per GPU (weak scaling: every rank
owns the index tile [rank*64M, (rank+1)*64M) and generates it in place, untimed).
Rank 0 prints ONE JSON line.

Launched without a rendezvous (`WORLD_SIZE` unset) and with --gpus N > 1, this file starts the N ranks itself —
as a child `python -m torch.distributed.run ... bench.py ...`, BEFORE anything in this process touches torch or
the GPU — relays rank 0's JSON line and exits with the child's return code.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B_STREAM = 76           # algorithmic HBM bytes per eval+sample unit: 32 in + 44 out
B_MAT = 4               # + the int32 material id of a mixed batch (80 B/unit in configs 4/5)
B_GATHER = 192          # algorithmic table bytes per unit: 2 lookups x 8 texels x 12 B (reported beside)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW")
HBM_COPY_GBS = 6290.0   # measured float4 copy on the same chip (same guide)
GATHER_CEILING_GBS = 7000.0   # whole-line fabric bytes of the bare memory pattern (tools/microbench/gather128.hip, gather_streams.hip: 6.9-7.0 TB/s)
SEED = 0x5EED
EXIT_GATHER_FAILED = 3  # the N>1 result-gather leg raised or ran into its deadline (the bench line is still printed)


def host_cores() -> int:
    """CPU threads this process may really use: affinity capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) // int(period))))
    except Exception:
        pass
    return n


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def measured_traffic(variant: int, layout: int, units: int, config: str):
    """Fabric (L2 <-> memory side) bytes per launch from the committed rocprofv3 PMC passes
    (profiles/traffic.json), if one exists for this kernel variant, table layout, batch size and config; else None."""
    try:
        rows = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["rows"]
        for r in rows:
            if (r["kernel_variant"] == min(variant, 3) and r["table_layout"] == layout and r["units"] == units
                    and r.get("config", "merl64m") == config):
                return r
    except Exception:
        pass
    return None


def measured_valu():
    """Executed VALU instructions per unit of the dominant kernel, from the committed SQ counter passes (profiles/valu.json)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "valu.json")))
    except Exception:
        return None


# SURVEY.md §8d "Flops" row: the VALU reading.  256 CU x 4 SIMD x 2.4 GHz; a wave-instruction occupies its SIMD for 4 cycles at
# the f64 / full-width rate (16 lanes per clock: 78.6 TFLOPS f64 FMA = 39.3 T lane-instructions/s) and for 2 at the plain-f32
# rate (157.3 TFLOPS f32 FMA = 78.6 T lane-instructions/s) — tools/microbench/valu_issue.hip, profiles/r03_valu_issue.json.
VALU_PEAK_T_LANE_INSTS_F64_RATE = 39.3
VALU_PEAK_T_LANE_INSTS_F32_RATE = 78.6


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--units", type=int, default=0, help="eval+sample units per GPU per step (0 = the config's size)")
    p.add_argument("--config", default="merl64m", choices=["merl64m", "ggx64m", "mixed16_256m", "resident100"],
                   help="merl64m = BASELINE configs[1] (the bench line); the others are the parity-test configs 3-5, "
                        "timed only on request: GGX alpha=0.1, 16 mixed MERL materials x 256M, 100 resident tables x (1B / 8) per GPU")
    p.add_argument("--table", default="ggx_tab", help="synthetic table kind, or a path to a real MERL .binary")
    p.add_argument("--lookup", choices=["trilinear", "nearest"], default="trilinear")
    p.add_argument("--kernel", type=int, default=-1, help="kernel variant (MRL_OPT_KERNEL); -1 = library default")
    p.add_argument("--layout", type=int, default=-1, help="table layout (MRL_OPT_TABLE_LAYOUT); -1 = library default")
    p.add_argument("--arena-mb", type=int, default=-1,
                   help="MRL_OPT_TABLE_ARENA_MB: place the tables back to back in one device allocation of this size; -1 (default): 20 GB for "
                        "--config resident100 and 4 GB for mixed16_256m (address translation bounds those launches; one arena removes the slow mode "
                        "of the process-to-process spread and is worth 5 %%: profiles/r03_arena_ab.txt), none otherwise; 0: one allocation per table")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-reps", type=int, default=5, help="repetitions of each CPU baseline leg (median is reported)")
    p.add_argument("--no-gather", action="store_true", help="N>1: skip the separately reported RCCL gather leg")
    p.add_argument("--gather-units", type=int, default=16 << 20, help="N>1: units per rank moved by the gather leg")
    p.add_argument("--gather-deadline", type=float, default=90.0, help="N>1: seconds after which the gather leg counts as failed")
    p.add_argument("--parity-sample", type=int, default=4096)
    p.add_argument("--coherent", type=int, default=0,
                   help="diagnostic, never the bench line: the inputs repeat with this period (units), so the table traffic is served by "
                        "L2 the way a real render's coherent rays are and the launch shows its non-fabric floor (VALU + LDS + streams)")
    p.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                   help="control-plane backend; gloo only rehearses the multi-rank logic (ranks may then share one GPU: --share-gpu)")
    p.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses GPU 0")
    p.add_argument("--launch-timeout", type=float, default=1500.0, help="self-launch: seconds before the child ranks are killed")
    p.add_argument("--no-scalar-calls", action="store_true", help="N=1: skip the separately reported one-unit call leg (lib/scalar_host)")
    p.add_argument("--no-native-group", action="store_true",
                   help="N>1: skip the separately reported run of the native C++ host (lib/group_host: one process, mrl_group over the N GPUs, RCCL gather)")
    p.add_argument("--native-units", type=int, default=8 << 20, help="N>1: units per device of the native C++ host's run")
    p.add_argument("--native-deadline", type=float, default=90.0, help="N>1: seconds before the native C++ host is killed")
    p.add_argument("--no-configs4", action="store_true",
                   help="N>1: skip the full-shape BASELINE configs[4] block (100 resident tables, --configs4-units pairs over the N devices, "
                        "compute-only / rgb-only gather / full gather for both transports, native C++ host)")
    p.add_argument("--configs4-units", type=int, default=10**9, help="N>1: total units of the configs[4] block (BASELINE: 1B pairs)")
    p.add_argument("--configs4-deadline", type=float, default=240.0, help="N>1: seconds before one leg of the configs[4] block is killed")
    p.add_argument("--configs4-reserve-cus", type=int, default=8,
                   help="N>1: compute units per device that the second run of each transport leaves to the transfer kernels (MRL_OPT_RESERVED_CUS)")
    return p.parse_args(argv)


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a rendezvous: start the N ranks as a child torch.distributed.run and
    relay rank 0's JSON line.  Nothing in THIS process imports torch or touches the GPU (no exec of a process
    that has initialised HIP; the ranks are ordinary children)."""
    import signal
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env, start_new_session=True)
    lines = []
    deadline = time.monotonic() + args.launch_timeout
    import threading

    def reader():
        for line in child.stdout:
            lines.append(line)

    t = threading.Thread(target=reader, daemon=True)
    t.start()
    timed_out = False
    while child.poll() is None:
        if time.monotonic() > deadline:
            timed_out = True
            try:
                os.killpg(child.pid, signal.SIGTERM)        # the exact process group this launcher started
                time.sleep(5.0)
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            break
        time.sleep(0.2)
    rc = child.wait()
    t.join(timeout=5.0)
    bench_line = None
    for line in lines:
        s = line.strip()
        if s.startswith("{") and '"metric"' in s:
            bench_line = s
        else:
            sys.stderr.write(line)
    if bench_line is not None:
        print(bench_line, flush=True)
    if timed_out:
        sys.stderr.write(f"bench.py: the {args.gpus} ranks did not finish within {args.launch_timeout} s and were killed\n")
        return rc or 124
    if bench_line is None and rc == 0:
        sys.stderr.write("bench.py: the ranks exited 0 but rank 0 printed no bench line\n")
        return 1
    return rc


def native_group_leg(n_gpus: int, share_gpu: bool, units: int, deadline: float) -> dict:
    """Schema: compute_only_Meval_s (no inter-device traffic), rgb_gathered_Meval_s (eval alone, 12 B/unit to
    the root), gathered_Meval_s (the fused unit, 44 B/unit to the root), root_ingress_GBps, transport ("rccl" | "peer_copy"),
    fallback_from (null, or what failed before the run was repeated with device copies in a fresh process), selftest
    (per-link GB/s of 1-64 MB payloads, bit-checked, before the pipeline), check_mismatches (must be 0).
    The native multi-device host path on the same GPUs, reported beside `value`: examples/group_host.cpp (plain C++
    over mrl_group_*, no Python in it) runs as a CHILD process after this job's ranks have released their GPUs —
    replicated tables, tiles generated in place, sharded eval+sample with the chunk-pipelined result gather (RCCL
    point-to-point when the devices are distinct), checked inside the program against a single-device run."""
    import subprocess
    exe = os.path.join(ROOT, "mitsuba_customization_amd", "lib", "group_host")
    if not os.path.exists(exe):
        return {"skipped": "lib/group_host is not built"}
    devices = ",".join("0" if share_gpu else str(i) for i in range(n_gpus))
    cmd = [exe, "--devices", devices, "--units-per-device", str(units), "--chunk", str(max(1, units // 4)),
           "--steps", "3", "--warmup", "1", "--check", "--selftest"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=deadline)
    except subprocess.TimeoutExpired:
        return {"failed": f"lib/group_host did not finish within {deadline} s and was killed", "cmd": " ".join(cmd)}
    except Exception as e:
        return {"failed": repr(e), "cmd": " ".join(cmd)}
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not line:
        return {"failed": f"exit code {r.returncode}", "stderr": r.stderr[-400:], "cmd": " ".join(cmd)}
    out = json.loads(line[-1])
    out["cmd"] = " ".join(cmd)
    return out


def configs4_block(n_gpus: int, share_gpu: bool, total_units: int, deadline: float, reserve_cus: int) -> dict:
    """BASELINE configs[4] at its full shape, beside `value` (never in it): ALL 100 tables resident on every device, `total_units`
    pairs (1e9) in N index tiles, the chunk-pipelined gather of every result to device 0 — through lib/group_host (plain C++ over
    mrl_group_*; one process, one RCCL communicator per device).  One leg per (transport, reserved CUs): each a fresh process under its
    own deadline, so that a transport that hangs or dies costs its leg, not the others.  Per leg (DESIGN.md §7 has the schema):
    compute_only_Meval_s (nothing crosses a link), rgb_gathered_Meval_s (eval alone: 12 B/unit to the root), gathered_Meval_s (the fused
    unit: 44 B/unit), root_ingress_GBps, check_mismatches (the gathered arrays against one device evaluating the whole range: must be 0),
    selftest (per-link GB/s, bit-checked; a failure is reported, not fatal).  reserved_cus > 0 answers what nobody could test on one
    GPU: RCCL's send / receive are KERNELS and the compute grids are persistent — do the transfers of chunk k overlap the compute of
    chunk k + 1, or wait for a grid to drain?  The second run of a transport keeps `reserve_cus` CUs per device free for them
    (MRL_OPT_RESERVED_CUS: a CU-masked compute stream, verified with hardware ids in profiles/r04_cu_mask_probe.json)."""
    import subprocess
    exe = os.path.join(ROOT, "mitsuba_customization_amd", "lib", "group_host")
    if not os.path.exists(exe):
        return {"skipped": "lib/group_host is not built"}
    devices = ",".join("0" if share_gpu else str(i) for i in range(n_gpus))
    per_device = (total_units + n_gpus - 1) // n_gpus
    chunk = max(1, min(8 << 20, per_device // 4))
    transports = ["copy"] if share_gpu else ["rccl", "copy"]       # RCCL refuses two ranks on one device: the rehearsal runs copies only
    legs = []
    for transport in transports:
        for reserve in ((0, reserve_cus) if reserve_cus > 0 else (0,)):
            cmd = [exe, "--devices", devices, "--transport", transport, "--no-fallback", "--tables", "100", "--units-per-device", str(per_device),
                   "--chunk", str(chunk), "--steps", "2", "--warmup", "1", "--check", "--selftest", "--reserve-cus", str(reserve)]
            leg = {"transport_asked": transport, "reserved_cus": reserve, "cmd": " ".join(cmd)}
            t0 = time.time()
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=deadline)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                if r.returncode != 0 or not line:
                    leg["failed"] = f"exit code {r.returncode}"
                    leg["stderr"] = r.stderr[-600:]
                else:
                    leg.update(json.loads(line[-1]))
            except subprocess.TimeoutExpired:
                leg["failed"] = f"did not finish within {deadline} s and was killed"
            except Exception as e:
                leg["failed"] = repr(e)
            leg["wall_s"] = round(time.time() - t0, 1)
            legs.append(leg)
    ok = [g for g in legs if "failed" not in g]
    return {"workload": f"BASELINE configs[4]: 100 MERL tables resident per device, {total_units} pairs tile-sharded over {n_gpus} device(s), "
                        "results gathered to device 0 in pipelined chunks", "total_units": total_units, "units_per_device": per_device,
            "chunk_units": chunk, "legs": legs, "legs_ok": len(ok), "all_legs_failed": not ok}


def scalar_calls_leg(deadline: float = 60.0) -> dict:
    """What ONE-unit calls cost (the virtual BSDF::eval / sample / pdf of a stock per-ray integrator; mrl_scalar_*),
    reported beside `value`: examples/scalar_host.cpp as a child process once this process has released the GPU — microseconds
    per call from one thread and amortised over 16, every answer bit-compared with the batch call inside the program."""
    import subprocess
    exe = os.path.join(ROOT, "mitsuba_customization_amd", "lib", "scalar_host")
    if not os.path.exists(exe):
        return {"skipped": "lib/scalar_host is not built"}
    cmd = [exe, "--threads", "16", "--calls", "5000"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=deadline)
    except subprocess.TimeoutExpired:
        return {"failed": f"lib/scalar_host did not finish within {deadline} s and was killed"}
    except Exception as e:
        return {"failed": repr(e)}
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not line:
        return {"failed": f"exit code {r.returncode}", "stderr": r.stderr[-300:]}
    d = json.loads(line[-1])
    return {"us_per_call_one_thread": d["solo_us_per_call"], "us_per_eval_pdf_call_one_thread": d["solo_eval_pdf_us"],
            "us_per_sample_call_one_thread": d["solo_sample_us"], "us_per_call_amortised_16_threads": d["all_threads_us_per_call_amortised"],
            "answers_differing_from_the_batch_call": d["wrong"],
            # the plugins' default (scalar="cpu"): the same fused unit on the calling CPU thread (mrl_host_eval_sample)
            "cpu_path_us_per_call": d.get("cpu_path_us_per_call"), "cpu_path_us_per_call_amortised_16_threads": d.get("cpu_path_us_per_call_amortised"),
            "cpu_path_units_bit_identical_to_batch": d.get("cpu_path_units_bit_identical_to_batch"),
            "cpu_path_worst_rel_diff_to_batch": d.get("cpu_path_worst_rel_diff_to_batch"), "cmd": " ".join(cmd)}


def cpu_baseline(ob, table, lookup: int, reps: int) -> dict:
    """the scalar f64 oracle behind a Mitsuba-0.6-style virtual call on the GPU box's host cores:
    N = 2^20 units on one thread (BASELINE configs[0]'s size) and N = 16 * 2^20 units on all threads, median of `reps`."""
    cores = host_cores()
    opts = ob.make_opts(lookup=lookup)
    n1, nN = 1 << 20, 16 << 20
    t1 = sorted(ob.bench_merl(table, n1, 1, SEED, True, opts)[0] for _ in range(reps))
    tN = sorted(ob.bench_merl(table, nN, cores, SEED, True, opts)[0] for _ in range(reps))
    e1 = sorted(ob.bench_merl(table, n1, 1, SEED, False, opts)[0] for _ in range(max(1, reps // 2 + 1)))
    m1, mN, me = t1[len(t1) // 2], tN[len(tN) // 2], e1[len(e1) // 2]
    return {
        "value": round(nN / mN / 1e6, 4),
        "unit": "Meval/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{nN} eval+sample units of the same workload (pair indices 0..{nN - 1}) on {cores} threads, "
                  f"and {n1} units (BASELINE configs[0]'s size) on 1 thread; scalar f64 oracle behind a "
                  f"Mitsuba-0.6-style virtual call, median of {reps} repetitions each",
        "single_thread_value": round(n1 / m1 / 1e6, 4),
        "single_thread_eval_only_value": round(n1 / me / 1e6, 4),
        "single_thread_units": n1,
        "repetitions": reps,
        "cpu_model": cpu_model(),
    }


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))                         # before torch / the GPU are touched in this process

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hot path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} needs GPU {local_rank}, the box has {torch.cuda.device_count()} "
                 "(--share-gpu with --dist-backend gloo rehearses the multi-rank logic on one GPU)")
    torch.cuda.set_device(local_rank)
    use_pg = "RANK" in os.environ and "MASTER_PORT" in os.environ      # launched by torch.distributed.run
    if world > 1 and not use_pg:
        sys.exit("bench.py: WORLD_SIZE > 1 without a rendezvous (use torch.distributed.run)")
    if use_pg:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(minutes=10))
        else:
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(minutes=10))
    comm_dev = (lambda d: d) if args.dist_backend == "nccl" else (lambda d: torch.device("cpu"))

    from mitsuba_customization_amd import host, synth

    gpu = host.MerlHip(local_rank)
    gpu.use_torch_stream()
    if args.kernel >= 0:
        gpu.set_option(host.OPT_KERNEL, args.kernel)
    gpu.set_option(host.OPT_LOOKUP, 1 if args.lookup == "trilinear" else 0)
    if args.layout >= 0:
        gpu.set_option(host.OPT_TABLE_LAYOUT, args.layout)
    elif args.lookup == "nearest":
        gpu.set_option(host.OPT_TABLE_LAYOUT, host.LAYOUT_ROWS)     # one texel per lookup: the compact layout wins

    if args.arena_mb < 0:
        args.arena_mb = {"resident100": 20480, "mixed16_256m": 4096}.get(args.config, 0)
    if args.arena_mb:
        gpu.set_option(host.OPT_TABLE_ARENA_MB, args.arena_mb)
    GGX = (0.1, (0.143, 0.375, 1.442), (3.983, 2.386, 1.603))      # BASELINE config 3: alpha 0.1, gold-like eta / k
    n_tables = {"merl64m": 1, "ggx64m": 0, "mixed16_256m": 16, "resident100": 100}[args.config]
    default_units = {"merl64m": 64 << 20, "ggx64m": 64 << 20, "mixed16_256m": 256 << 20, "resident100": 125_000_000}[args.config]
    workload = {"merl64m": "BASELINE configs[1]: single MERL material, 64M (wi,wo,u) batched eval+sample per GPU",
                "ggx64m": "BASELINE configs[2]: GGX rough conductor alpha=0.1, 64M pairs per GPU (not the bench line)",
                "mixed16_256m": "BASELINE configs[3]: 16 MERL materials mixed in one batch, 256M pairs per GPU (not the bench line)",
                "resident100": "BASELINE configs[4]: 100 MERL tables resident, 1B pairs / 8 = 125M per GPU (not the bench line)"}[args.config]
    tables = []
    if os.path.exists(args.table) and n_tables == 1:
        tables = [synth.read_merl_binary(args.table)]
        table_name = os.path.basename(args.table)
    else:
        # distinct synthetic tables up to 16, then cycled (each upload is its own resident copy in HBM)
        distinct = [synth.make_table(args.table, s) for s in range(min(n_tables, 16))]
        tables = [distinct[i % len(distinct)] for i in range(n_tables)]
        table_name = f"synthetic {args.table} seeds 0..{max(0, min(n_tables, 16) - 1)} (MERL layout; no real MERL file offline)"
    table = tables[0] if tables else None
    ids = [gpu.upload_merl(t) for t in tables]
    if args.config == "ggx64m":
        ids = [gpu.ggx(*GGX)]
        table_name = "analytic GGX, no table"
    mid = ids[0]

    n = args.units or default_units
    first = rank * n
    wi, wo, u = gpu.generate_pairs(SEED, first, n)          # untimed, in place on the device
    dev = wi.device
    if args.coherent and args.coherent < n:
        p_ = args.coherent
        reps = (n + p_ - 1) // p_
        wi, wo, u = (x[:p_].repeat(reps, 1)[:n].contiguous() for x in (wi, wo, u))
    mat = None
    if len(ids) > 1:
        mat = gpu.generate_materials(SEED, first, n, len(ids))
        mat += ids[0]
    out = (torch.empty((n, 3), dtype=torch.float32, device=dev), torch.empty((n,), dtype=torch.float32, device=dev),
           torch.empty((n, 3), dtype=torch.float32, device=dev), torch.empty((n,), dtype=torch.float32, device=dev),
           torch.empty((n, 3), dtype=torch.float32, device=dev))

    def step():
        gpu.eval_sample(wi, wo, u, mat=mat, material=mid, out=out)

    def fence():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    gpu.timer_start()                                       # hipEvents on the launch stream (= torch's current)
    for _ in range(args.steps):
        step()
    kernel_ms = gpu.timer_stop() / max(args.steps, 1)       # avg launch duration of the dominant kernel
    fence()
    elapsed = time.perf_counter() - t0
    if use_pg:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=comm_dev(dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    b_unit = B_STREAM + (B_MAT if mat is not None else 0)
    total_units = float(n) * world * args.steps
    value = total_units / elapsed / 1e6                     # M eval+sample units / s, whole job
    achieved = b_unit * n / (kernel_ms * 1e-3) / 1e9        # GB/s of algorithmic stream bytes, one launch on one GPU
    achieved_g = (b_unit + B_GATHER) * n / (kernel_ms * 1e-3) / 1e9
    variant, layout = gpu.get_option(host.OPT_KERNEL), gpu.get_option(host.OPT_TABLE_LAYOUT)
    kname = {0: "k_batch<eval_sample>", 1: "k_table<eval_sample>", 2: "k_table<eval_sample,nt>"}.get(variant, "k_table_dma<eval_sample>")
    if args.config == "ggx64m":
        kname = "k_ggx<eval_sample>" if variant >= 1 else "k_batch<eval_sample>"
    if variant >= 3 and (layout != 1 or args.lookup != "trilinear"):
        kname = "k_table<eval_sample,nt>"
    traffic = measured_traffic(variant, layout, n, args.config)
    mem = gpu.memory_info()
    lib_sources = host.build_info()                        # "sources <hash>": what the running library was built from

    roofline = {
        "bound": "hbm",
        "achieved": round(achieved, 2),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5),
        # bytes per launch that crossed the L2 <-> memory-side fabric (TCC_EA0 read/write requests); Infinity-Cache
        # hits are counted in them, so this is an upper bound on HBM bytes (MI355X_MICROARCH.md, HBM section)
        "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
        "traffic_source": traffic["source"] if traffic else None,
        # the counters are not re-collected in this run (PMC passes need rocprofv3 around the process): they come from
        # profiles/traffic.json, which records the library sources they were measured on — stale when this library differs
        "traffic_measured_on": traffic.get("library", "unrecorded (before round 3)") if traffic else None,
        "traffic_stale": (traffic.get("library") != lib_sources) if traffic else None,
        "kernel": kname,
        "kernel_ms": round(kernel_ms, 4),
        "bytes_per_unit": b_unit,
        "frac_of_measured_copy_peak": round(achieved / HBM_COPY_GBS, 5),
        "with_gather": {"bytes_per_unit": b_unit + B_GATHER, "achieved": round(achieved_g, 2),
                        "frac": round(achieved_g / HBM_PEAK_GBS, 5)},
        "note": "achieved/frac count the algorithmic STREAM bytes only (32 B in + 44 B out per unit, + 4 B material id in "
                "mixed batches; SURVEY.md §8d).  With the brick layout a table is 187 MB and misses L2, so the 2 x 8-texel "
                "gather (192 B/unit algorithmic, 256 B/unit fetched as two 128-B lines) crosses the fabric too: with_gather "
                "prices stream + gather bytes against the same peak",
    }
    valu = measured_valu() if args.config == "merl64m" else None
    if valu:
        lane_insts = valu["valu_insts_per_unit"] * n / (kernel_ms * 1e-3) / 1e12
        roofline["valu"] = {
            "what": "SURVEY.md §8d 'Flops' row: executed VALU instructions per unit (one lane = one unit; SQ_INSTS_VALU per wave-iteration) x units/s, "
                    "against the chip's vector issue rate — not the bound of this launch on random inputs (the fabric is), the bound on cache-served ones",
            "valu_insts_per_unit": valu["valu_insts_per_unit"], "f64_arith_per_unit": valu["f64_fma_mul_add_per_unit"],
            "f64_transcendental_per_unit": valu["f64_transcendental_per_unit"], "convert_per_unit": valu["convert_per_unit"],
            "T_lane_insts_per_s": round(lane_insts, 2),
            "frac_of_f64_rate_peak": round(lane_insts / VALU_PEAK_T_LANE_INSTS_F64_RATE, 4),
            "frac_of_f32_rate_peak": round(lane_insts / VALU_PEAK_T_LANE_INSTS_F32_RATE, 4),
            "peaks_T_lane_insts_per_s": {"f64_rate (78.6 TFLOPS f64 FMA / 2)": VALU_PEAK_T_LANE_INSTS_F64_RATE, "f32_rate (157.3 TFLOPS f32 FMA / 2)": VALU_PEAK_T_LANE_INSTS_F32_RATE},
            "measured_on": valu.get("library"), "stale": valu.get("library") != lib_sources, "source": valu.get("source"),
        }
    if traffic:
        gbps = traffic["hbm_bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9
        roofline["fabric_traffic"] = {
            "what": "TCC_EA0 read + write request bytes per launch / this run's kernel time: requests served by the "
                    "Infinity Cache are included, so this is fabric (EA) traffic, not proven HBM traffic",
            "GBps": round(gbps, 1),
            "frac_of_random_line_gather_ceiling": round(gbps / GATHER_CEILING_GBS, 4),
            "frac_of_measured_copy_peak": round(gbps / HBM_COPY_GBS, 4),
            "frac_of_hbm_spec_peak": round(gbps / HBM_PEAK_GBS, 4),
            "over_algorithmic_stream_bytes": round(traffic["hbm_bytes_per_launch"] / (b_unit * n), 3),
        }

    result = {
        "metric": "bsdf_eval_sample_throughput",
        "value": round(value, 3),
        "unit": "Meval/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": workload,
            "materials_resident": len(ids),
            "resident_table_bytes": mem["table_bytes"],
            "units_per_gpu_per_step": n,
            "table": table_name,
            "lookup": args.lookup,
            "kernel_variant": variant,
            "table_layout": layout,
            "library": lib_sources,
            **({"table_arena_mb": args.arena_mb} if args.arena_mb else {}),
            "sharding": f"index tiles x{world}, tables replicated, no data-path collective",
            **({"coherent_period": args.coherent, "not_the_bench_line": "inputs repeat: table traffic is L2-served"} if args.coherent else {}),
        },
        "roofline": roofline,
    }

    # ---- rank 0, N=1: parity sample vs the oracle + CPU baseline on the host cores ----
    if rank == 0:
        from oracle import binding as ob                       # checker / cpu_baseline leg only
        k = min(args.parity_sample, n)
        if k > 0:
            idx = (torch.arange(k, device=dev, dtype=torch.int64) * (n - 1)) // max(k - 1, 1)   # exact integers (f32 linspace rounds n-1 up to n)
            hin = [x[idx].cpu().numpy() for x in (wi, wo, u)]
            hout = [x[idx].cpu().numpy() for x in out]
            lookup = 1 if args.lookup == "trilinear" else 0
            if args.config == "ggx64m":
                G = ob.OracleGgx(float(np.float32(GGX[0])), [float(np.float32(x)) for x in GGX[1]], [float(np.float32(x)) for x in GGX[2]])
                s_wo, s_pdf, s_w = G.sample(hin[0], hin[2])
                ref = (G.eval(hin[0], hin[1]), G.pdf(hin[0], hin[1]), s_wo, s_pdf, s_w)
            else:
                hm = None if mat is None else (mat[idx] - ids[0]).cpu().numpy()
                ref = ob.eval_sample_multi([ob.OracleTable(t) for t in tables], hin[0], hin[1], hin[2], hm, ob.make_opts(lookup=lookup))
            worst, beyond = 0.0, 0
            for k_out, (got, want) in enumerate(zip(hout, ref)):
                got = got.astype(np.float64); want = want.astype(np.float64)
                err = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
                # GGX sampled directions are f64 results rounded to f32: one ulp (1.2e-7 absolute) may flip
                slack = 1.2e-7 if (args.config == "ggx64m" and k_out == 2) else 0.0
                err = np.where(np.abs(got - want) <= 1e-30 + slack, 0.0, err)
                worst = max(worst, float(err.max()))
                beyond += int((err > 1e-6).sum())
            result["parity"] = {"sample": k, "max_rel_err_vs_oracle": worst, "values_beyond_tolerance": beyond,
                                "tolerance": 1e-6, "pinned": False}
            if lookup == 0:
                result["parity"]["note"] = ("nearest lookup: a coordinate on an exact bin edge may land in the neighbouring texel "
                                            "(values_beyond_tolerance counts those flips)")
        if world == 1 and not args.no_cpu_baseline and args.config == "merl64m":
            result["cpu_baseline"] = cpu_baseline(ob, table, 1 if args.lookup == "trilinear" else 0, max(1, args.cpu_reps))

    # ---- N>1: the RCCL result gather, reported beside (never inside) `value` ----
    # It runs LAST and under a deadline.  If it raises on a rank or stalls, rank 0 still prints the bench line with
    # "gather": {"failed": reason, "phase": ...} and EVERY rank exits with EXIT_GATHER_FAILED (non-zero): a process that
    # has GPU work in flight behind a dead peer is not retried and not reported as a clean run.
    exit_code = 0
    if world > 1 and not args.no_gather:
        import threading
        finished = threading.Event()
        phase = {"name": "alloc"}

        def give_up():
            if finished.is_set():
                return
            why = f"rank {rank}: gather leg still in phase '{phase['name']}' after {args.gather_deadline} s"
            sys.stderr.write("bench.py: " + why + "\n")
            if rank == 0:
                result["gather"] = {"failed": why, "phase": phase["name"]}
                print(json.dumps(result), flush=True)
            os._exit(EXIT_GATHER_FAILED)

        watchdog = threading.Timer(args.gather_deadline, give_up)
        watchdog.daemon = True
        watchdog.start()
        # bounded: the first <= 16M units of every rank's outputs (704 MB per rank); every rank first agrees
        # that its buffers exist, so that a failed allocation on one rank skips the leg everywhere
        # instead of leaving the others waiting in a send
        g_units = min(n, args.gather_units)
        ok, full, err = 1, None, ""
        try:
            from mitsuba_customization_amd import shard
            local = [o[:g_units].to(comm_dev(dev)) for o in out]       # gloo rehearsal: host copies
            if rank == 0:
                full = [torch.empty((g_units * world,) + tuple(t.shape[1:]), dtype=t.dtype, device=comm_dev(dev)) for t in local]
        except Exception as e:
            ok, err = 0, repr(e)
        try:
            phase["name"] = "all_reduce(buffers allocated)"
            flag = torch.tensor([ok], dtype=torch.int32, device=comm_dev(dev))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 1:
                phase["name"] = "batch_isend_irecv"
                g = shard.bench_gather(local, steps=3, out=full)
                g["units_per_rank"] = g_units
                g["extrapolated_ms_for_full_step"] = round(g["ms"] * n / g_units, 3)
            else:
                g = {"skipped": err or "another rank could not allocate its buffers"}     # agreed by every rank: clean
        except Exception as e:                                          # the peers run into their own deadline
            g = {"failed": repr(e), "phase": phase["name"]}
            sys.stderr.write(f"bench.py: rank {rank}: gather leg raised in phase '{phase['name']}': {e!r}\n")
            exit_code = EXIT_GATHER_FAILED
        del full
        if rank == 0:
            result["gather"] = g
        finished.set()
        watchdog.cancel()

    printed = {"done": False}

    def emit():
        if rank == 0 and not printed["done"]:
            printed["done"] = True
            print(json.dumps(result), flush=True)

    if use_pg:
        if exit_code:
            emit()
            os._exit(exit_code)                             # peers may be stuck in the leg: no collective teardown
        # teardown under a deadline as well; a stall here is a failure, not a clean run
        import threading

        def teardown_stalled():
            sys.stderr.write(f"bench.py: rank {rank}: teardown (barrier / destroy_process_group) stalled for 60 s\n")
            result["teardown"] = {"failed": "barrier / destroy_process_group stalled for 60 s"}
            emit()
            os._exit(EXIT_GATHER_FAILED)

        bye = threading.Timer(60.0, teardown_stalled)
        bye.daemon = True
        bye.start()
        gpu.close()
        dist.barrier()
        dist.destroy_process_group()
        bye.cancel()
    else:
        gpu.close()

    # ---- N>1: the native C++ host over the same GPUs, once every rank has let go of them; beside `value`, never in it ----
    if rank == 0 and world > 1 and not args.no_native_group:
        del wi, wo, u, out, mat
        torch.cuda.empty_cache()
        result["native_group"] = native_group_leg(world, args.share_gpu, args.native_units, args.native_deadline)
        if not args.no_configs4:
            result["configs4"] = configs4_block(world, args.share_gpu, args.configs4_units, args.configs4_deadline, args.configs4_reserve_cus)
            configs4_dead = bool(result["configs4"].get("all_legs_failed"))
    # ---- N=1: the per-ray plugin path (one-unit calls), beside `value`, never in it ----
    if world == 1 and args.config == "merl64m" and not args.no_scalar_calls and not args.no_cpu_baseline:
        result["scalar_calls"] = scalar_calls_leg()
    emit()
    if rank == 0 and world > 1 and locals().get("configs4_dead"):
        sys.stderr.write("bench.py: every leg of the configs[4] block failed\n")
        sys.exit(EXIT_GATHER_FAILED)                             # the line above is complete; the exit code says the block is not


if __name__ == "__main__":
    main()
