"""f16mx: the same episodes through different chunk plans (different GEMM tile shapes) must give the same bits.
python tools/mx_invariance.py [precision]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims

prec = sys.argv[1] if len(sys.argv) > 1 else "f16mx"
E, A, K, T = 64, 5, 20, 12
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 23), joint=True, step=4)
eng.set_tuning("lanes", lanes)
g = torch.Generator().manual_seed(3)
ctx = torch.randn([E, A, 256], generator=g).cuda()
x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
exact = eng.denoise(x_T, ctx, None, precision="f32", want_pos=False)[0].clone()
ref = None
for chunk in (0, 64, 51, 32, 17, 8, 5, 2, 1):
    eng.set_chunk_episodes(chunk)
    v = eng.denoise(x_T, ctx, None, precision=prec, want_pos=False)[0].clone()
    d = (v - exact).norm(dim=-1).mean().item()
    if ref is None:
        ref = v
    nd = int((v != ref).any(dim=-1).reshape(E, -1).any(dim=1).sum())
    print(f"[{prec}] chunk={chunk}: mean ADE vs f32 {d:.3e}; episodes differing from the first plan: {nd}", flush=True)
