"""Multi-GPU path without GPUs: tile arithmetic, and the result gather
over torch.distributed with the gloo backend at world_size 2 and 3.  The per-tile "compute" is the
CPU oracle on a synthetic table, so the root can check the gathered arrays against a single-process
run bit for bit (results are a pure function of the unit index)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mitsuba_customization_amd import shard


def test_tile_bounds_partition_everything_once():
    for n in (0, 1, 7, 64, 1000, 10**9 + 7):
        for world in (1, 2, 3, 4, 8):
            tiles = [shard.tile_bounds(n, world, r) for r in range(world)]
            assert tiles[0][0] == 0 and tiles[-1][1] == n
            for (a, b), (c, d) in zip(tiles, tiles[1:]):
                assert b == c and a <= b and c <= d
            per = -(-n // world) if n else 0
            assert all(b - a <= per for a, b in tiles)
    with pytest.raises(ValueError):
        shard.tile_bounds(10, 2, 2)
    assert shard.tile_bounds(1_000_000_000, 8, 7) == (875_000_000, 1_000_000_000)   # BASELINE config 5 split


def test_chunk_ranges():
    assert shard.chunk_ranges(10, 25, 10) == [(10, 20), (20, 25)]
    assert shard.chunk_ranges(5, 5, 3) == []
    assert sum(b - a for a, b in shard.chunk_ranges(0, 1001, 64)) == 1001


def test_single_process_run_is_identity():
    def compute(lo, hi):
        i = torch.arange(lo, hi, dtype=torch.float32)
        return [torch.stack([i, 2 * i, 3 * i], 1), i + 0.5]
    a, b = shard.run_sharded(compute, 1000, 64, gather=True)
    assert torch.equal(b, torch.arange(1000, dtype=torch.float32) + 0.5) and a.shape == (1000, 3)
    assert shard.run_sharded(compute, 0, 64) == []


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_compute(seed, table):
    from oracle import binding as ob
    T = ob.OracleTable(table)

    def compute(lo, hi):
        wi, wo, u = ob.generate_pairs(seed, lo, hi - lo)
        outs = ob.eval_sample_multi([T], wi, wo, u, None)
        return [torch.from_numpy(np.ascontiguousarray(o)) for o in outs]
    return compute


def _worker(rank, world, port, n_total, chunk, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mitsuba_customization_amd import synth
        table = synth.make_table("ggx_tab", 1, (12, 10, 16))          # small table: the oracle only needs dims
        compute = _oracle_compute(0x5EED, table)
        # 1) chunk-pipelined run with gather to rank 0
        full = shard.run_sharded(compute, n_total, chunk, gather=True, dst=0)
        # 2) one-shot gather of whole tiles
        lo, hi = shard.tile_bounds(n_total, world, rank)
        tile = compute(lo, hi) if hi > lo else [torch.empty((0, 3)), torch.empty((0,)), torch.empty((0, 3)), torch.empty((0,)), torch.empty((0, 3))]
        full2 = shard.gather_tiles(tile, n_total, dst=0)
        # 3) no gather: every rank keeps its tile
        mine = shard.run_sharded(compute, n_total, chunk, gather=False)
        assert all(m.shape[0] == hi - lo for m in mine)
        if rank == 0:
            ref = compute(0, n_total)
            ok = all(torch.equal(a, b) for a, b in zip(full, ref)) and all(torch.equal(a, b) for a, b in zip(full2, ref))
            ok = ok and all(torch.equal(m, r[lo:hi]) for m, r in zip(mine, ref))
            with open(result_path, "w") as f:
                f.write("ok" if ok else "mismatch")
        else:
            assert full is None and full2 is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,chunk", [(2, 5000, 700), (2, 4096, 4096), (3, 1001, 128)])
def test_gloo_gather_is_bit_identical_to_single_process(tmp_path, world, n_total, chunk):
    from oracle import binding as ob
    ob.build()
    result = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, _free_port(), n_total, chunk, result), nprocs=world, join=True)
    assert open(result).read() == "ok"


def _edge_worker(rank, world, port, result_path):
    """(a) the root owns an EMPTY tile (n_total < world, dst != 0): it still has to allocate the full arrays;
    (b) a non-default group: tiles are numbered by group rank, messages addressed by global rank."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def compute(lo, hi):
            i = torch.arange(lo, hi, dtype=torch.float32)
            return [torch.stack([i, 2 * i, 3 * i], 1), (i + 0.5).to(torch.float64)]
        ok = True
        # (a) 2 units over 3 ranks, gathered to rank 2 whose tile is empty
        full = shard.run_sharded(compute, 2, 64, gather=True, dst=2)
        if rank == 2:
            ref = compute(0, 2)
            ok = ok and full is not None and all(torch.equal(a, b) and a.dtype == b.dtype for a, b in zip(full, ref))
        else:
            ok = ok and full is None
        # nothing at all
        assert shard.run_sharded(compute, 0, 64, gather=True, dst=1) in ([], None)
        # (b) sub-group {1, 2}: group rank 0 = global rank 1 is the root
        sub = dist.new_group(ranks=[1, 2])
        if rank in (1, 2):
            full = shard.run_sharded(compute, 777, 100, gather=True, dst=0, group=sub)
            tile = compute(*shard.tile_bounds(777, 2, rank - 1))
            full2 = shard.gather_tiles(tile, 777, dst=0, group=sub)
            if rank == 1:
                ref = compute(0, 777)
                ok = ok and all(torch.equal(a, b) for a, b in zip(full, ref)) and all(torch.equal(a, b) for a, b in zip(full2, ref))
            else:
                ok = ok and full is None and full2 is None
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            open(result_path, "w").write("ok" if int(flag) == 1 else "mismatch")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gloo_empty_root_tile_and_subgroup(tmp_path):
    result = str(tmp_path / "edge.txt")
    mp.spawn(_edge_worker, args=(3, _free_port(), result), nprocs=3, join=True)
    assert open(result).read() == "ok"
