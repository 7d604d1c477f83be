import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
E, A, K, T = 64, 5, 20, 12
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 23), joint=True, step=4)
g = torch.Generator().manual_seed(3)
ctx = torch.randn([E, A, 256], generator=g).cuda()
x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
eng.set_tuning("lanes", 1)
eng.set_chunk_episodes(64)
ref = eng.denoise(x_T, ctx, None, precision="f16mx", want_pos=False)[0].clone()
for lanes in (1, 2):
    for knobs in ({}, {"ln_rows": 128}, {"ln_rows": 64}, {"ln_fuse": 2}, {"gemm_h_variant": 6}):
        eng.set_tuning("lanes", lanes)
        for k in ("ln_rows", "ln_fuse", "gemm_h_variant"):
            eng.set_tuning(k, 0)
        for k, v in knobs.items():
            eng.set_tuning(k, v)
        eng.set_chunk_episodes(32)
        for rep in range(2):
            v = eng.denoise(x_T, ctx, None, precision="f16mx", want_pos=False)[0].clone()
            eps = (v != ref).any(dim=-1).reshape(E, -1).any(dim=1).nonzero().flatten().tolist()
            print(f"lanes={lanes} {knobs} rep {rep}: {len(eps)} episodes differ: {eps[:40]}  max {float((v - ref).abs().max()):.2e}", flush=True)
