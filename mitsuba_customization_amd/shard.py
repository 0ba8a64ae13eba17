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


def _peer(group, r: int) -> int:
    """P2POp addresses peers by GLOBAL rank; tiles are numbered by group rank."""
    return dist.get_global_rank(group, r) if group is not None else r


def _backend_moves_device_tensors(group=None) -> bool:
    """RCCL ("nccl") moves device tensors itself; gloo's point-to-point only takes host memory."""
    return dist.get_backend(group) == "nccl"


class _Posted:
    """One chunk's point-to-point messages in flight.  `tensors` keeps the buffers alive until wait()."""

    def __init__(self, works, tensors, after=None):
        self.works, self.tensors, self.after = works, tensors, after

    def wait(self):
        for w in self.works:
            w.wait()
        if self.after is not None:
            self.after()
        self.tensors = None


def _post(specs, group, comm_stream=None, ready=None) -> _Posted:
    """Posts one group of point-to-point messages.  specs: [("send" | "recv", tensor, group rank of the peer)].

    Host tensors, or device tensors over RCCL: the tensors go to batch_isend_irecv as they are (one grouped
    ncclSend/ncclRecv set: every peer uses its own xGMI link to the root), issued on `comm_stream` behind the `ready`
    event when given, so the messages of chunk k overlap the compute of chunk k+1.
    Device tensors over a host-only backend (gloo rehearsal, ranks sharing one GPU): staged through pinned host
    buffers on the same side stream — send: D2H copy behind `ready`, then the host message; recv: host message,
    then an H2D copy into the destination slice when the message is waited for."""
    if not specs:
        return _Posted([], [])
    device = specs[0][1].is_cuda
    if not device or _backend_moves_device_tensors(group):
        ops = [dist.P2POp(dist.isend if kind == "send" else dist.irecv, t, _peer(group, peer), group) for kind, t, peer in specs]
        if device and comm_stream is not None:
            with torch.cuda.stream(comm_stream):
                if ready is not None:
                    comm_stream.wait_event(ready)
                return _Posted(dist.batch_isend_irecv(ops), [t for _, t, _ in specs])
        return _Posted(dist.batch_isend_irecv(ops), [t for _, t, _ in specs])
    # staged path
    stream = comm_stream if comm_stream is not None else torch.cuda.current_stream()
    staged, ops = [], []
    with torch.cuda.stream(stream):
        if ready is not None:
            stream.wait_event(ready)
        for kind, t, peer in specs:
            h = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
            if kind == "send":
                h.copy_(t, non_blocking=True)
            staged.append((kind, t, h))
            ops.append(dist.P2POp(dist.isend if kind == "send" else dist.irecv, h, _peer(group, peer), group))
        copied = torch.cuda.Event()
        copied.record(stream)
    copied.synchronize()                                     # the host message reads the pinned buffers

    def land():
        with torch.cuda.stream(stream):
            for kind, t, h in staged:
                if kind == "recv":
                    t.copy_(h, non_blocking=True)
            done = torch.cuda.Event()
            done.record(stream)
        done.synchronize()                                   # pinned buffers may be dropped after this

    return _Posted(dist.batch_isend_irecv(ops), staged, after=land)


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
        specs = []
        for peer in range(world):
            plo, phi = tile_bounds(n_total, world, peer)
            if phi == plo:
                continue
            for k, t in enumerate(local):
                if peer == dst:
                    full[k][plo:phi].copy_(t)
                else:
                    specs.append(("recv", full[k][plo:phi], peer))
        _post(specs, group).wait()
        return full
    if hi > lo:
        _post([("send", t.contiguous(), dst) for t in local], group).wait()
    return None


def _agree_on_layout(outs, n_total: int, world: int, dst: int, group) -> List[tuple]:
    """(trailing shape, dtype, device) of every output array, known on the root even when its own tile is empty
    (n_total < world with dst != 0): group rank 0 always owns a non-empty tile and tells everyone."""
    dlo, dhi = tile_bounds(n_total, world, dst)
    if dhi > dlo:
        return [(tuple(t.shape[1:]), t.dtype, t.device) for t in outs] if outs is not None else []
    meta = [[(tuple(t.shape[1:]), str(t.dtype).split(".")[-1], t.is_cuda) for t in outs]] if outs is not None else [None]
    dist.broadcast_object_list(meta, src=_peer(group, 0), group=group)
    here = torch.device("cuda", torch.cuda.current_device()) if (meta[0] and meta[0][0][2]) else torch.device("cpu")
    return [(tuple(tail), getattr(torch, dt), here) for tail, dt, _ in meta[0]]


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
    pending: List[_Posted] = []
    comm_stream = None

    def drain(keep: int):
        while len(pending) > keep:
            pending.pop(0).wait()

    for c in range(steps):
        a, b = min(hi, lo + c * chunk), min(hi, lo + (c + 1) * chunk)
        outs = [t for t in compute(a, b)] if b > a else None
        if not gather or world == 1:
            if outs is not None:
                tile_out.append(outs)
            continue
        if c == 0:
            layout = _agree_on_layout(outs, n_total, world, dst, group)
            if is_root:
                full = [torch.empty((n_total,) + tail, dtype=dt, device=dev) for tail, dt, dev in layout]
            if overlap and layout and layout[0][2].type == "cuda":
                comm_stream = torch.cuda.Stream()
        ready = None
        if comm_stream is not None and outs is not None:
            ready = torch.cuda.Event()
            ready.record()
        if is_root:
            specs = []
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
                        specs.append(("recv", full[k][pa:pb], peer))
            if specs:
                pending.append(_post(specs, group, comm_stream, ready))
        elif outs is not None:
            pending.append(_post([("send", t.contiguous(), dst) for t in outs], group, comm_stream, ready))
        drain(keep=1)         # at most one chunk of messages in flight behind the compute
    drain(keep=0)
    if comm_stream is not None:
        torch.cuda.current_stream().wait_stream(comm_stream)

    if not gather or world == 1:
        if not tile_out:
            return []
        return [torch.cat([o[k] for o in tile_out], dim=0) for k in range(len(tile_out[0]))]
    if is_root and full is None:
        return []                                            # n_total == 0
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
