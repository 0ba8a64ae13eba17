"""Device-pointer calls are plain kernel launches on the caller's stream, so a run of them can be captured into a
HIP graph and replayed (small batches are launch-bound: tools/launch_overhead.py)."""
import pytest

pytestmark = pytest.mark.gpu


def test_calls_capture_into_a_hip_graph_and_replay_identically():
    import torch
    from mitsuba_customization_amd import host, synth
    with host.MerlHip(0) as g:
        t = g.upload_merl(synth.make_table("ggx_tab", seed=2))
        gg = g.ggx(0.1, (0.2, 0.9, 1.1), (3.9, 2.4, 2.2))
        n = 1 << 12
        wi, wo, u = g.generate_pairs(3, 0, n)
        mat = (torch.arange(n, device="cuda") % 2).to(torch.int32) * gg + t * 0
        want = [o.clone() for o in g.eval_sample(wi, wo, u, mat=mat)]
        queue = torch.arange(0, n, 2, dtype=torch.int32, device="cuda")
        count = torch.tensor([queue.numel()], dtype=torch.int32, device="cuda")
        want_q = g.eval_queue(wi, wo, queue, count, material=t).clone()
        out = tuple(torch.zeros_like(o) for o in want)
        out_q = torch.zeros_like(want_q)
        torch.cuda.synchronize()
        graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        with torch.cuda.graph(graph, stream=side):
            g.eval_sample(wi, wo, u, mat=mat, out=out)
            g.eval_queue(wi, wo, queue, count, material=t, out=out_q)
        for o in out:
            o.zero_()
        out_q.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out, want)) and torch.equal(out_q, want_q)
        # new inputs in the same buffers, and a new device-side queue length, are picked up by a replay
        wi2, wo2, u2 = g.generate_pairs(4, 0, n)
        wi.copy_(wi2); wo.copy_(wo2); u.copy_(u2); count.fill_(100)
        out_q.zero_()
        graph.replay()
        torch.cuda.synchronize()
        fresh = g.eval_sample(wi, wo, u, mat=mat)
        assert all(torch.equal(a, b) for a, b in zip(out, fresh))
        assert int((out_q.abs().sum(-1) > 0).sum()) <= 100


def test_rgl_calls_capture_too_alone_and_inside_a_mixed_batch():
    """An RGL material's launch, and the two launches of a batch that mixes it with a table and an analytic material, replay from a graph."""
    import torch
    from mitsuba_customization_amd import host, synth
    with host.MerlHip(0) as g:
        t = g.upload_table(synth.make_table("noise", 5, (8, 8, 12)), (1.0, 1.0, 1.0))
        gg = g.ggx(0.1, (0.2, 0.9, 1.1), (3.9, 2.4, 2.2))
        r = g.upload_rgl(synth.make_rgl_fields(seed=17, n_phi=1, n_theta=4, res=8))
        n = 1 << 12
        wi, wo, u = g.generate_pairs(5, 0, n)
        ids = torch.tensor([t, gg, r], device="cuda", dtype=torch.int32)
        mat = ids[torch.arange(n, device="cuda") % 3]
        want_mixed = [o.clone() for o in g.eval_sample(wi, wo, u, mat=mat)]
        want_alone = [o.clone() for o in g.eval_sample(wi, wo, u, material=r)]
        out_mixed = tuple(torch.zeros_like(o) for o in want_mixed)
        out_alone = tuple(torch.zeros_like(o) for o in want_alone)
        torch.cuda.synchronize()
        graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        with torch.cuda.graph(graph, stream=side):
            g.eval_sample(wi, wo, u, mat=mat, out=out_mixed)
            g.eval_sample(wi, wo, u, material=r, out=out_alone)
        for o in out_mixed + out_alone:
            o.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out_mixed, want_mixed)) and all(torch.equal(a, b) for a, b in zip(out_alone, want_alone))
        assert float(out_mixed[0][2::3].abs().max()) > 0                       # the RGL units of the mixed batch were evaluated
