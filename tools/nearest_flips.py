import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from mitsuba_customization_amd import host, synth
from oracle import binding as ob
n = 1 << 24
for kind, seed in (("ggx_tab", 0), ("noise", 5)):
    tab = synth.make_table(kind, seed)
    with host.MerlHip(0) as g:
        g.set_option(host.OPT_LOOKUP, 0); g.set_option(host.OPT_TABLE_LAYOUT, 0)
        mid = g.upload_merl(tab)
        wi, wo, u = g.generate_pairs(0x5EED, 0, n)
        got = [t.cpu().numpy() for t in g.eval_sample(wi, wo, u, material=mid)]
        hin = [t.cpu().numpy() for t in (wi, wo, u)]
    want = ob.eval_sample_multi([ob.OracleTable(tab)], *hin, None, ob.make_opts(lookup=0))
    for k, name in ((0, "rgb"), (4, "weight")):
        bad = (np.abs(got[k].astype(np.float64) - want[k]) > 1e-6 * np.abs(want[k]) + 1e-30).any(axis=1)
        print(kind, name, "units with a value beyond 1e-6:", int(bad.sum()), "of", n, flush=True)
