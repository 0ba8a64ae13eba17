import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from mitsuba_customization_amd import host, synth
n = 1 << 22
with host.MerlHip(0) as g:
    rgb = g.upload_merl(synth.make_table("ggx_tab", 0))
    wi, wo, u = g.generate_pairs(0x5EED, 0, n)
    for v in (1, 3):
        g.set_option(host.OPT_KERNEL, v)
        fused = [t.clone() for t in g.eval_sample(wi, wo, u, material=rgb)]
        wo2, pdf2, w = [t.clone() for t in g.sample(wi, u, material=rgb)]
        f2 = g.eval(wi, wo2, material=rgb).clone()
        print("variant", v, "sample-only vs fused weights mismatch:", int((w != fused[4]).sum()), " wo2:", int((wo2 != fused[2]).sum()))
        q = f2 / pdf2[:, None]
        bad = (w != q)
        print("   weight != eval/pdf:", int(bad.sum()))
        f2d, pd, wd = f2.double(), pdf2.double(), w.double()
        for i in bad.nonzero()[:6]:
            r, c = i.tolist()
            exact = f2d[r, c] / pd[r]
            print("     unit", r, "ch", c, "f2", float(f2[r, c]).hex(), "pdf", float(pdf2[r]).hex(), "w", float(w[r, c]).hex(), "torch", float(q[r, c]).hex(),
                  "exact64", float(exact).hex(), "w*pdf", float(wd[r, c] * pd[r]).hex())
