"""mrl_partition_by_material: the wavefront ballot/prefix compaction primitive.  The partition must be stable
(ascending slot indices inside each material's group), complete, drop ids that name no material, and feed the queue
entry points: per-material queue calls over the partition reproduce the mixed whole-array call bit for bit."""
import pytest

pytestmark = pytest.mark.gpu


def _check_partition(mat, k, queue, offsets, counts):
    import torch
    off = offsets.cpu().tolist()
    assert off[0] == 0 and counts.cpu().tolist() == [off[m + 1] - off[m] for m in range(k)]
    assert off[k] == int(((mat >= 0) & (mat < k)).sum())
    for m in range(k):
        want = (mat == m).nonzero().flatten().to(torch.int32)
        assert torch.equal(queue[off[m]:off[m + 1]], want), m


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1_000_003, 8_000_000])
@pytest.mark.parametrize("k", [1, 5, 100])
def test_partition_is_stable_and_complete(n, k):
    import torch
    from mitsuba_customization_amd import host
    with host.MerlHip(0) as g:
        for i in range(k):
            g.ggx(0.05 + 0.001 * i, (0.2, 0.9, 1.1), (3.9, 2.4, 2.2))
        gen = torch.Generator(device="cuda").manual_seed(n * 131 + k)
        mat = torch.randint(-1, k + 1, (n,), device="cuda", generator=gen, dtype=torch.int32)    # -1 and k name no material
        if n > 1000:
            mat[100:400] = 0                                                                       # long uniform run
        queue, offsets, counts = g.partition_by_material(mat)
        g.synchronize()
        _check_partition(mat, k, queue, offsets, counts)


def test_per_material_queue_calls_reproduce_the_mixed_batch():
    import torch
    from mitsuba_customization_amd import host, synth
    n = 1 << 18
    with host.MerlHip(0) as g:
        ids = [g.upload_merl(synth.make_table("ggx_tab", seed=s)) for s in range(3)] + [g.ggx(0.1, (0.2, 0.9, 1.1), (3.9, 2.4, 2.2))]
        wi, wo, u = g.generate_pairs(9, 0, n)
        mat = g.generate_materials(9, 0, n, len(ids))
        want = g.eval_sample(wi, wo, u, mat=mat)
        queue, offsets, counts = g.partition_by_material(mat)
        off = offsets.cpu().tolist()                                   # the one read-back: where each group starts
        got = tuple(torch.full_like(t, -3.0) for t in want)
        for m in ids:
            g.eval_sample_queue(wi, wo, u, queue[off[m]:off[m + 1]], counts[m:m + 1], material=m, out=got)
        g.synchronize()
        for a, b in zip(got, want):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_partition_rejects_host_pointers_and_too_many_slots():
    import numpy as np
    from mitsuba_customization_amd import host
    with host.MerlHip(0) as g:
        g.ggx(0.1, (0.2, 0.9, 1.1), (3.9, 2.4, 2.2))
        m = np.zeros(8, np.int32); q = np.zeros(8, np.uint32); o = np.zeros(2, np.uint32); c = np.zeros(1, np.uint32)
        L = g._lib
        assert L.mrl_partition_by_material(g._ctx, m.ctypes.data, 8, q.ctypes.data, o.ctypes.data, c.ctypes.data) == -7
        assert L.mrl_partition_by_material(g._ctx, None, 8, None, None, None) == -1
