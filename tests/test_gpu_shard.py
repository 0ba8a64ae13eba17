"""Rehearsal of the sharded path on the 1-GPU box: 2 ranks share GPU 0 for the compute, the control
plane and the result gather run over gloo with host copies (RCCL needs one GPU per rank).  Checks what
the multi-GPU run relies on: tiles generated in place from the pair index + gathered results are
bit-identical to a single-process run over the whole range."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, chunk, result_path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mitsuba_customization_amd import host, shard, synth
        gpu = host.MerlHip(0)
        mid = gpu.upload_merl(synth.make_table("ggx_tab", 0))

        def compute(lo, hi):                       # inputs generated in place from the pair index
            wi, wo, u = gpu.generate_pairs(0x5EED, lo, hi - lo)
            return [t.cpu() for t in gpu.eval_sample(wi, wo, u, material=mid)]

        full = shard.run_sharded(compute, n_total, chunk, gather=True, dst=0)
        if rank == 0:
            ref = compute(0, n_total)
            ok = all(torch.equal(a, b) for a, b in zip(full, ref))
            open(result_path, "w").write("ok" if ok else "mismatch")
        dist.barrier()
        gpu.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_sharing_the_gpu_reproduce_the_single_process_run(tmp_path):
    import torch.multiprocessing as mp
    result = str(tmp_path / "r.txt")
    mp.spawn(_worker, args=(2, _free_port(), 300_001, 70_000, result), nprocs=2, join=True)
    assert open(result).read() == "ok"


def _device_worker(rank, world, port, n_total, chunk, result_path):
    """Device tensors all the way: compute() returns GPU arrays, so run_sharded takes its overlap branch — a side
    stream that waits on the chunk's `ready` event — and, because gloo cannot move device memory (and RCCL refuses two
    ranks on one GPU), the messages are staged through pinned host buffers on that side stream."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mitsuba_customization_amd import host, shard, synth
        torch.cuda.set_device(0)
        gpu = host.MerlHip(0)
        mid = gpu.upload_merl(synth.make_table("ggx_tab", 0))
        streams_seen = set()
        real_stream = torch.cuda.Stream

        class SpyStream(real_stream):                   # counts the side streams run_sharded creates
            def __new__(cls, *a, **k):
                s = super().__new__(cls, *a, **k)
                streams_seen.add(s.cuda_stream)
                return s
        torch.cuda.Stream = SpyStream

        def compute(lo, hi):
            wi, wo, u = gpu.generate_pairs(0x5EED, lo, hi - lo)
            return list(gpu.eval_sample(wi, wo, u, material=mid))          # device tensors

        full = shard.run_sharded(compute, n_total, chunk, gather=True, dst=1)
        torch.cuda.Stream = real_stream
        streams_seen.discard(torch.cuda.current_stream().cuda_stream)      # current_stream() also builds Stream objects
        tile = compute(*shard.tile_bounds(n_total, world, rank))
        full2 = shard.gather_tiles(tile, n_total, dst=1)
        if rank == 1:
            torch.cuda.synchronize()
            ref = compute(0, n_total)
            ok = all(a.is_cuda and torch.equal(a, b) for a, b in zip(full, ref)) and all(torch.equal(a, b) for a, b in zip(full2, ref))
            ok = ok and len(streams_seen) == 1
            open(result_path, "w").write("ok" if ok else "mismatch")
        else:
            assert full is None and full2 is None and len(streams_seen) == 1
        dist.barrier()
        gpu.close()
    finally:
        dist.destroy_process_group()


def test_device_tensors_take_the_side_stream_overlap_path(tmp_path):
    import torch.multiprocessing as mp
    result = str(tmp_path / "d.txt")
    mp.spawn(_device_worker, args=(2, _free_port(), 250_003, 60_000, result), nprocs=2, join=True)
    assert open(result).read() == "ok"
