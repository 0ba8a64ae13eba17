"""Device groups without a GPU: the tile / chunk arithmetic the native multi-device path shares with shard.py
(pure functions exported by the library), and the failure mode of mrl_group_init on a box without a GPU."""
import ctypes as C

import pytest

from mitsuba_customization_amd import host, shard


def test_tile_bounds_match_the_python_sharder():
    for n in (0, 1, 7, 64, 1000, 12345, 10**9, 10**9 + 7):
        for world in (1, 2, 3, 4, 8):
            tiles = [host.tile_bounds(n, world, r) for r in range(world)]
            assert tiles == [shard.tile_bounds(n, world, r) for r in range(world)]
            assert tiles[0][0] == 0 and tiles[-1][1] == n
    assert host.tile_bounds(10, 2, 2) == (0, 0) and host.tile_bounds(10, 0, 0) == (0, 0)        # out of range: empty, no abort
    assert host.tile_bounds(1_000_000_000, 8, 7) == (875_000_000, 1_000_000_000)               # BASELINE config 5


def test_chunks_cover_each_tile_once_in_order():
    for n, world, chunk in ((1001, 3, 128), (5000, 2, 700), (4096, 2, 4096), (7, 8, 3), (10**9, 8, 8 << 20)):
        steps = host.chunk_steps(n, world, chunk)
        per = -(-n // world)
        assert steps == -(-per // chunk)
        for r in range(world):
            lo, hi = host.tile_bounds(n, world, r)
            at = lo
            for c in range(steps):
                a, b = host.chunk_bounds(n, world, r, chunk, c)
                if b > a:
                    assert a == at and b - a <= chunk and b <= hi
                    at = b
                else:
                    assert at == hi                                       # exhausted tiles give empty chunks
            assert at == hi
            assert host.chunk_bounds(n, world, r, chunk, steps) == (hi, hi)
            # the same schedule shard.run_sharded walks
            assert [host.chunk_bounds(n, world, r, chunk, c) for c in range(steps) if host.chunk_bounds(n, world, r, chunk, c)[1] > host.chunk_bounds(n, world, r, chunk, c)[0]] \
                == shard.chunk_ranges(lo, hi, chunk)
    assert host.chunk_steps(0, 4, 64) == 0 and host.chunk_steps(64, 4, 0) == 0


def test_group_init_argument_checks_and_no_device():
    import torch
    L = host.load_library()
    g = C.c_void_p()
    ids = (C.c_int * 2)(0, 0)
    assert L.mrl_group_init(0, ids, 0, C.byref(g)) == host.ERR_INVALID
    assert L.mrl_group_init(2, None, 0, C.byref(g)) == host.ERR_INVALID
    assert L.mrl_group_init(2, ids, 7, C.byref(g)) == host.ERR_INVALID
    assert L.mrl_group_init(2, ids, host.TRANSPORT_RCCL, C.byref(g)) == host.ERR_INVALID       # RCCL needs distinct devices
    assert L.mrl_group_size(None) == host.ERR_INVALID and L.mrl_group_destroy(None) == 0
    if not torch.cuda.is_available():
        assert L.mrl_group_init(1, ids, 0, C.byref(g)) == host.ERR_NO_DEVICE and not g.value   # no CPU fallback
        assert b"no gfx950" in L.mrl_group_last_error(None)
        with pytest.raises(host.MerlHipError):
            host.MerlGroup([0])


# ------------------------------------------------------------------ the pipeline's bookkeeping against a stub transport
class _StubGroup:
    """Executes mrl_group_plan's operations the way sharded() does, with numpy arrays for device memory and a transport
    whose transfers complete LATE: a transfer only reads its source buffer when it is retired, and retiring happens either
    when a later compute has to wait for it (`after_transfer_of_step` / a previous call's transfer out of the same buffer —
    the sent[] events) or at the end of the call.  A compute that overwrites a buffer with an unretired transfer in flight
    is the bug the double-buffering must prevent; the stub raises on it."""

    def __init__(self, world, cap):
        import numpy as np
        self.np = np
        self.world, self.cap = world, cap
        self.buf = [[np.full(cap, -1, np.int64), np.full(cap, -1, np.int64)] for _ in range(world)]
        self.in_flight = {}                     # (member, buffer) -> (first, count): posted, not yet retired
        self.sent_valid = [[False, False] for _ in range(world)]

    def retire(self, member, buffer, out):
        first, count = self.in_flight.pop((member, buffer))
        out[first:first + count] = self.buf[member][buffer][:count]

    def run(self, n_total, chunk, root, out):
        np = self.np
        for op in host.group_plan(n_total, self.world, chunk, root):
            lo, _hi = host.tile_bounds(n_total, self.world, op.member)
            assert op.first - lo == op.tile_offset and op.count > 0
            units = np.arange(op.first, op.first + op.count, dtype=np.int64)          # "eval" of a unit = its index
            if op.kind == host.PLAN_COMPUTE:
                if op.buffer < 0:
                    assert op.member == root
                    out[op.first:op.first + op.count] = units
                    continue
                assert op.member != root and op.count <= self.cap, "a chunk exceeds the member's buffer"
                key = (op.member, op.buffer)
                if self.sent_valid[op.member][op.buffer] and key in self.in_flight:
                    self.retire(op.member, op.buffer, out)                             # hipStreamWaitEvent(compute, sent[s])
                assert key not in self.in_flight, f"step {op.step}: member {op.member} overwrites buffer {op.buffer} while its transfer is in flight"
                if op.after_transfer_of_step >= 0:
                    assert op.after_transfer_of_step == op.step - 2
                self.buf[op.member][op.buffer][:op.count] = units
            else:
                assert op.member != root and op.buffer in (0, 1)
                self.in_flight[(op.member, op.buffer)] = (op.first, op.count)
                self.sent_valid[op.member][op.buffer] = True
        # (the call returns here; its last transfers are still in flight — the next call's first steps must wait for them)

    def drain(self, out):
        for member, buffer in list(self.in_flight):
            self.retire(member, buffer, out)


@pytest.mark.parametrize("n,world,chunk,root", [(1001, 3, 128, 0), (5000, 2, 700, 1), (4096, 4, 4096, 2), (5, 4, 2, 3), (7, 8, 3, 0),
                                                (100_000, 8, 1000, 5), (64, 1, 16, 0), (999, 5, 1, 4)])
def test_plan_gathers_every_unit_once_through_a_stub_transport(n, world, chunk, root):
    import numpy as np
    plan = host.group_plan(n, world, chunk, root)
    steps = host.chunk_steps(n, world, chunk)
    assert plan == [] if n == 0 else len(plan) > 0
    # issue order: steps ascend; inside a step every compute precedes every transfer; one compute per member and step
    order = [(op.step, op.kind) for op in plan]
    assert order == sorted(order)
    assert len({(op.step, op.member, op.kind) for op in plan}) == len(plan)
    assert max(op.step for op in plan) == steps - 1
    computed = np.zeros(n, np.int64)
    for op in plan:
        if op.kind == host.PLAN_COMPUTE:
            computed[op.first:op.first + op.count] += 1
            a, b = host.chunk_bounds(n, world, op.member, chunk, op.step)
            assert (a, b) == (op.first, op.first + op.count)
            assert (op.buffer == -1) == (op.member == root) and (op.member == root or op.buffer == op.step % 2)
            assert op.after_transfer_of_step == (op.step - 2 if (op.member != root and op.step >= 2) else -1)
    assert (computed == 1).all(), "a unit is computed twice or not at all"
    transfers = sorted((op.member, op.step, op.buffer, op.first, op.count) for op in plan if op.kind == host.PLAN_TRANSFER)
    assert transfers == sorted((op.member, op.step, op.buffer, op.first, op.count) for op in plan if op.kind == host.PLAN_COMPUTE and op.member != root)
    # ... and executed against the late-completing stub, three calls in a row over the same buffers
    per = -(-n // world)
    g = _StubGroup(world, min(chunk, per))
    for call in range(3):
        out = np.full(n, -7, np.int64)
        g.run(n, chunk, root, out)              # raises if a compute overwrites a buffer whose transfer is still in flight
        if call == 2:
            g.drain(out)                        # mrl_group_synchronize
            assert (out == np.arange(n)).all()
        else:
            # the call has returned with its last transfers in flight: its arrays are complete once those retire, and the
            # NEXT call's first two steps must wait for them before reusing the buffers (sent_valid survives the call)
            done = out.copy()
            for (member, buffer), (first, count) in g.in_flight.items():
                done[first:first + count] = g.buf[member][buffer][:count]
            assert (done == np.arange(n)).all()


def test_plan_rejects_nonsense_and_sizes_itself():
    assert host.group_plan(0, 4, 16, 0) == [] and host.group_plan(100, 0, 16, 0) == [] and host.group_plan(100, 4, 0, 0) == []
    assert host.group_plan(100, 4, 16, 4) == [] and host.group_plan(100, 4, 16, -1) == []
    L = host.load_library()
    L.mrl_group_plan.restype = C.c_size_t
    L.mrl_group_plan.argtypes = [C.c_size_t, C.c_int, C.c_size_t, C.c_int, C.POINTER(host.PlanOp), C.c_size_t]
    ops = (host.PlanOp * 2)()
    n = L.mrl_group_plan(1000, 4, 100, 0, ops, 2)                # too small a buffer: the count is still the full one
    assert n == len(host.group_plan(1000, 4, 100, 0)) and n > 2
    assert hasattr(L, "mrl_group_link_test")


def test_stub_transport_catches_a_single_buffered_schedule(monkeypatch):
    """The stub is a real check: a schedule that always uses buffer 0 (no double buffering) overwrites a chunk in flight."""
    real = host.group_plan

    def single_buffered(n, world, chunk, root):
        ops = real(n, world, chunk, root)
        for op in ops:
            if op.buffer == 1:
                op.buffer = 0
        return ops

    monkeypatch.setattr(host, "group_plan", single_buffered)
    g = _StubGroup(2, 100)
    import numpy as np
    out = np.zeros(1000, np.int64)
    g.sent_valid = [[False, False], [False, False]]

    class NeverValid(list):                     # a transport that forgets to mark its sends: nothing ever waits
        def __getitem__(self, i):
            return [False, False]
    g.sent_valid = NeverValid()
    with pytest.raises((AssertionError, TypeError)):
        g.run(1000, 100, 0, out)
