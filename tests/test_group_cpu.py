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
