"""Index-tile sharding of a batch over the GPUs of one node + the result gather.

Every (wi, wo, u) unit is independent, so the path shards with NO data-path collective: rank r
owns the contiguous index tile [r*ceil(N/G), min(N, (r+1)*ceil(N/G))), generates / receives its
inputs in place, keeps every material table replicated, and computes its tile alone.  The only
communication is delivering per-tile RESULTS to a root that wants them in one place: peers send
their tile straight to the root (point-to-point send/recv in one group — each peer's own xGMI link
to the root, not a ring), placed directly into the root's full-size output arrays.  Long tiles are
cut into chunks so that the send of chunk k overlaps the compute of chunk k+1.

One process per GPU under torch.distributed (backend "nccl" = RCCL on the GPU box, "gloo" in the
CPU tests).  Results are a pure function of the unit index, hence bit-identical for any G.
"""
from __future__ import annotations

import time
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def tile_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous index tile of `rank` (may be empty for trailing ranks when n_total < world)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    per = -(-n_total // world) if n_total > 0 else 0
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


def chunk_ranges(lo: int, hi: int, chunk: int) -> List[Tuple[int, int]]:
    if chunk < 1:
        raise ValueError("chunk must be positive")
    return [(a, min(hi, a + chunk)) for a in range(lo, hi, chunk)]


def _world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def gather_tiles(local: Sequence[torch.Tensor], n_total: int, dst: int = 0, group=None,
                 out: Optional[Sequence[torch.Tensor]] = None) -> Optional[List[torch.Tensor]]:
    """Gathers per-rank tile arrays (first dim = units of the rank's tile, in tile_bounds order) to
    `dst`, which returns full arrays in global unit order; other ranks return None.  Peers send with
    one point-to-point message per array; the root receives in place into slices of the full arrays."""
    world, rank = _world(group)
    lo, hi = tile_bounds(n_total, world, rank)
    for t in local:
        if t.shape[0] != hi - lo:
            raise ValueError(f"rank {rank}: tile array has {t.shape[0]} units, tile is [{lo},{hi})")
    if world == 1:
        return [t for t in local]
    if rank == dst:
        full = list(out) if out is not None else [torch.empty((n_total,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in local]
        ops = []
        for peer in range(world):
            plo, phi = tile_bounds(n_total, world, peer)
            if phi == plo:
                continue
            for k, t in enumerate(local):
                if peer == dst:
                    full[k][plo:phi].copy_(t)
                else:
                    ops.append(dist.P2POp(dist.irecv, full[k][plo:phi], peer, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return full
    if hi > lo:
        ops = [dist.P2POp(dist.isend, t.contiguous(), dst, group) for t in local]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return None


def run_sharded(compute: Callable[[int, int], Sequence[torch.Tensor]], n_total: int, chunk: int,
                gather: bool = True, dst: int = 0, group=None, overlap: bool = True):
    """Runs `compute(lo, hi)` (returns the output arrays of units [lo, hi)) over this rank's tile in
    chunks of `chunk` units.  With gather=True the root ends up with full arrays in global order and
    returns them (peers return None); with gather=False every rank returns its tile's arrays.

    Pipelining: every rank walks the same chunk schedule (chunk c of every tile in step c), so in
    step c the root posts the receives for chunk c of all peers while everyone computes chunk c+1.
    On GPUs `compute` should enqueue on the current stream; the sends are issued from a side stream
    that waits on an event recorded after the chunk's compute."""
    world, rank = _world(group)
    lo, hi = tile_bounds(n_total, world, rank)
    per = -(-n_total // world) if n_total > 0 else 0
    steps = -(-per // chunk) if per > 0 else 0
    is_root = rank == dst
    full: Optional[List[torch.Tensor]] = None
    tile_out: List[List[torch.Tensor]] = []
    pending = []          # (works, keepalive tensors)
    use_cuda = False      # decided by the first chunk's tensors: overlap on a side stream only for device arrays
    comm_stream = None

    def drain(keep: int):
        while len(pending) > keep:
            works, _keep = pending.pop(0)
            for w in works:
                w.wait()

    for c in range(steps):
        a, b = min(hi, lo + c * chunk), min(hi, lo + (c + 1) * chunk)
        outs = [t for t in compute(a, b)] if b > a else None
        if outs is not None and comm_stream is None and overlap and world > 1 and gather and outs[0].is_cuda:
            use_cuda = True
            comm_stream = torch.cuda.Stream()
        if not gather or world == 1:
            if outs is not None:
                tile_out.append(outs)
            continue
        ready = None
        if use_cuda:
            ready = torch.cuda.Event()
            ready.record()
        if is_root:
            if full is None and outs is not None:
                full = [torch.empty((n_total,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in outs]
            ops = []
            for peer in range(world):
                plo, phi = tile_bounds(n_total, world, peer)
                pa, pb = min(phi, plo + c * chunk), min(phi, plo + (c + 1) * chunk)
                if pb <= pa:
                    continue
                if peer == dst:
                    for k, t in enumerate(outs):
                        full[k][pa:pb].copy_(t)
                else:
                    for k in range(len(full)):
                        ops.append(dist.P2POp(dist.irecv, full[k][pa:pb], peer, group))
            if ops:
                if use_cuda:
                    with torch.cuda.stream(comm_stream):
                        comm_stream.wait_event(ready)
                        pending.append((dist.batch_isend_irecv(ops), outs))
                else:
                    pending.append((dist.batch_isend_irecv(ops), outs))
        elif outs is not None:
            ops = [dist.P2POp(dist.isend, t.contiguous(), dst, group) for t in outs]
            if use_cuda:
                with torch.cuda.stream(comm_stream):
                    comm_stream.wait_event(ready)
                    pending.append((dist.batch_isend_irecv(ops), outs))
            else:
                pending.append((dist.batch_isend_irecv(ops), outs))
        drain(keep=1)         # at most one chunk of messages in flight behind the compute
    drain(keep=0)
    if use_cuda:
        torch.cuda.current_stream().wait_stream(comm_stream)

    if not gather or world == 1:
        if not tile_out:
            return []
        return [torch.cat([o[k] for o in tile_out], dim=0) for k in range(len(tile_out[0]))]
    return full if is_root else None


def bench_gather(local: Sequence[torch.Tensor], steps: int = 3, dst: int = 0, group=None,
                 out: Optional[Sequence[torch.Tensor]] = None) -> dict:
    """Times the result gather of one step's outputs (all ranks hold equal tiles) — reported by
    bench.py beside the compute throughput, never inside it."""
    world, rank = _world(group)
    n_local = int(local[0].shape[0])
    n_total = n_local * world
    nbytes = sum(t.element_size() * t.numel() for t in local)
    full = list(out) if (out is not None and rank == dst) else None
    if rank == dst and full is None:
        full = [torch.empty((n_total,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in local]
    gather_tiles(local, n_total, dst, group, out=full)          # warm-up: connection set-up
    times = []
    for _ in range(steps):
        if world > 1:
            dist.barrier(group)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        gather_tiles(local, n_total, dst, group, out=full)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier(group)
        times.append(time.perf_counter() - t0)
    best = min(times)
    return {
        "what": "gather of one step's per-tile outputs (rgb, pdf, wo', pdf', weight') to rank 0, point-to-point, not overlapped",
        "bytes_into_root": nbytes * (world - 1),
        "ms": round(best * 1e3, 3),
        "root_ingress_GBps": round(nbytes * (world - 1) / best / 1e9, 2) if best > 0 else None,
    }
