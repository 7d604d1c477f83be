import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
E, A, K, T = 32, 5, 20, 12
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 23), joint=True, step=4)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    eng.set_tuning(k, int(v))
g = torch.Generator().manual_seed(3)
ctx = torch.randn([E, A, 256], generator=g).cuda()
x = torch.randn([E, K * A, T, 2], generator=g).cuda()
outs = [eng.net_eval(x, ctx, 0, precision="f16mx").clone() for _ in range(4)]
for i in range(1, 4):
    d = (outs[i] != outs[0]).any(dim=-1)          # [E, KA, T]
    eps = d.reshape(E, -1).any(dim=1).nonzero().flatten().tolist()
    tok = d.reshape(-1).nonzero().flatten()
    print(f"run {i}: {int(d.sum())} tokens differ, episodes {eps}, max {float((outs[i] - outs[0]).abs().max()):.2e}")
    if len(tok):
        t = tok.cpu().numpy()
        print("   token idx //64 (first 40 distinct tiles):", sorted(set((t // 64).tolist()))[:40], " n tiles", len(set((t // 64).tolist())))
