"""256 episodes in F16MX, two chunks in flight, under forced chunk sizes 16 ... 43 against the automatic plan: ms per 50-step call.
python tools/chunk_fine.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
prec = os.environ.get("PREC", "f16mx"); E = int(os.environ.get("EPISODES", "256")); A, K, T = 5, 20, 12
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 23), joint=True, step=50)
g = torch.Generator().manual_seed(3)
ctx = torch.randn([E, A, 256], generator=g).cuda()
x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
eng.set_tuning("lanes", int(os.environ.get("LANES", "2")))
for chunk in [int(c) for c in os.environ.get("CHUNKS", "0,16,17,20,21,22,24,25,26,28,29,32,34,37,43,0").split(",")]:
    eng.set_chunk_episodes(chunk)
    v = eng.denoise(x_T, ctx, None, precision=prec, want_pos=False)[0]
    eng.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        v = eng.denoise(x_T, ctx, None, precision=prec, want_pos=False)[0]
    eng.synchronize(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 4 * 1e3
    print(f"chunk={chunk}: {ms:.1f} ms ({E * A * K / ms:.2f} k)", flush=True)
