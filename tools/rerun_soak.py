"""Bit-reproducibility soak: N full 50-step calls on the same inputs, every output compared bit for bit with the first
call of the default path (one chunk in flight).   python tools/rerun_soak.py [precision] [calls] [episodes] [lanes]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
E = int(sys.argv[3]) if len(sys.argv) > 3 else 103
A, K, T = 5, 20, 12
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 23), joint=True, step=50)
g = torch.Generator().manual_seed(3)
ctx = torch.randn([E, A, 256], generator=g).cuda()
x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 1
chunk = int(sys.argv[5]) if len(sys.argv) > 5 else 0
for kv in sys.argv[6:]:                      # extra tuning knobs: key=value
    k, v = kv.split("=")
    eng.set_tuning(k, int(v))
eng.set_chunk_episodes(chunk)
eng.set_tuning("lanes", 1)
ref, bad = eng.denoise(x_T, ctx, None, precision=prec, want_pos=False)[0].clone(), 0    # lanes = 1 reference
eng.set_tuning("lanes", lanes)
for i in range(calls):
    v = eng.denoise(x_T, ctx, None, precision=prec, want_pos=False)[0]
    eng.synchronize()
    if not torch.equal(v, ref):
        bad += 1
        dd = (v - ref).norm(dim=-1)
        eps = (dd.reshape(E, -1).max(dim=1)[0] > 0).nonzero().flatten().tolist()
        print(f"call {i}: {int((v != ref).any(dim=-1).sum())} points differ, max {float(dd.max()):.2e} mean {float(dd.mean()):.2e}, "
              f"{len(eps)} episodes: {eps[:12]}", flush=True)
print(f"{prec} lanes={lanes} chunk={chunk} {sys.argv[6:]}: {calls} calls of {E} episodes ({E * A * K} trajectories each), {bad} differ from the first")
