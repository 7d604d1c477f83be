"""256 episodes under different chunk sizes x chunks in flight: ms per 50-step call.  python tools/chunk_lanes_sweep.py [precision] [E]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
prec = sys.argv[1] if len(sys.argv) > 1 else "f16mx"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 256
A, K, T = 5, 20, 12
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 23), joint=True, step=50)
g = torch.Generator().manual_seed(3)
ctx = torch.randn([E, A, 256], generator=g).cuda()
x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
ref = None
for chunk in ([int(c) for c in os.environ["CHUNKS"].split(",")] if os.environ.get("CHUNKS") else (0, 17, 21, 26, 32, 34, 43, 51)):
    row = []
    for lanes in (1, 2, 3):
        eng.set_tuning("lanes", lanes)
        eng.set_chunk_episodes(chunk)
        v = eng.denoise(x_T, ctx, None, precision=prec, want_pos=False)[0]
        eng.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            v = eng.denoise(x_T, ctx, None, precision=prec, want_pos=False)[0]
        eng.synchronize(); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        if ref is None:
            ref = v.clone()
        row.append(f"lanes {lanes}: {ms:.1f} ms ({E * A * K / ms:.2f} k){'' if torch.equal(v, ref) else ' DIFFERENT'}")
    print(f"[{prec}] E={E} chunk={chunk}: " + " | ".join(row), flush=True)
