import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine, JmidError
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
A, K, T = 5, 20, 12
w = JMIDWeights.from_seed(NetDims(ctx_dim=256), 23)
for E in (1, 2, 5, 8, 13, 17, 32):
    for knobs in ({}, {"gemm_h_variant": 3}, {"gemm_h_variant": 4}, {"gemm_h_variant": 5}, {"gemm_h_variant": 6}, {"ln_fuse": 2}):
        eng = JmidEngine(w, joint=True, step=4)
        for k, v in knobs.items():
            eng.set_tuning(k, v)
        g = torch.Generator().manual_seed(3)
        ctx = torch.randn([E, A, 256], generator=g).cuda()
        x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
        a = eng.denoise(x_T, ctx, None, precision="f16x2", want_pos=False)[0].clone()
        try:
            b = eng.denoise(x_T, ctx, None, precision="f16mx", want_pos=False)[0].clone()
            print(f"E={E} {knobs}: max |mx - x2| = {float((a - b).abs().max()):.3e}", flush=True)
        except JmidError as e:
            print(f"E={E} {knobs}: {e}", flush=True)
        eng.close()
