import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from mitsuba_customization_amd import host, synth
from oracle.binding import OracleRgl
case=dict(seed=3, n_phi=1, n_theta=1, res=2, res_ndf=2, res_sigma=2)
fields=synth.make_rgl_fields(**case); orc=OracleRgl(fields)
n=1<<15
with host.MerlHip(0) as g:
    mid=g.upload_rgl(fields)
    wi_t,wo_t,u_t=g.generate_pairs(0x861+3,0,n)
    wi,u=wi_t.cpu().numpy(),u_t.cpu().numpy()
    wo2,pdf2,w=(t.cpu().numpy() for t in g.sample(wi_t,u_t,material=mid))
    o_wo2,o_pdf2,o_w=orc.sample(wi,u)
    both=(o_pdf2>0)&(pdf2>0)
    c_rgb,c_pdf=orc.eval_pdf(wi[both],wo2[both])
    err=np.abs(pdf2[both]-c_pdf)/np.abs(c_pdf)
    bad=np.argsort(-err)[:5]
    idx=np.nonzero(both)[0][bad]
    np.set_printoptions(precision=9)
    for i,b in zip(idx,bad):
        d=wi[i].astype(np.float64); d/=np.linalg.norm(d)
        o=wo2[i].astype(np.float64); o/=np.linalg.norm(o)
        m=d+o; m/=np.linalg.norm(m)
        print(i, err[b], "wi_hex", [float(x).hex() for x in wi[i]], "wo_hex", [float(x).hex() for x in wo2[i]], "wi",wi[i],"wo2",wo2[i],"o_wo2",o_wo2[i],"pdf2",pdf2[i],"orc@wo2",c_pdf[b],"o_pdf2",o_pdf2[i], "phi_i",np.arctan2(d[1],d[0]),"phi_m",np.arctan2(m[1],m[0]), "theta_i", np.arccos(d[2]))
