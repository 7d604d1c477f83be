import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 23), joint=True, step=4)
E, A, K, T = 32, 5, 20, 12
g = torch.Generator().manual_seed(3)
ctx = torch.randn([E, A, 256], generator=g); x_T = torch.randn([E, K * A, T, 2], generator=g)
perm = torch.randperm(E, generator=g)
def run():
    v, _ = eng.denoise(x_T.cuda(), ctx.cuda(), precision="f16x3", want_pos=False)
    vp, _ = eng.denoise(x_T[perm].cuda(), ctx[perm].cuda(), precision="f16x3", want_pos=False)
    v2, _ = eng.denoise(x_T.cuda(), ctx.cuda(), precision="f16x3", want_pos=False)
    v, vp, v2 = v.cpu(), vp.cpu(), v2.cpu()
    d = (vp - v[perm]).abs()
    bad_eps = [(int(perm[i]), i) for i in range(E) if d[i].max() > 0]
    return float((v - v2).abs().max()), float(d.max()), int((d > 0).sum()), bad_eps
for gv, av in [(0, 0), (2, 0), (1, 0), (0, 1), (2, 1), (1, 1)]:
    eng.set_tuning("gemm_h_variant", gv); eng.set_tuning("attn_h_variant", av)
    rr, mx, cnt, bad = run()
    print(f"gemm_v={gv} attn_v={av}: rerun maxdiff={rr:.3e}  perm maxdiff={mx:.3e} n_diff={cnt} episodes(orig,newpos)={bad[:12]}", flush=True)
