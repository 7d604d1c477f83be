"""Mid-size batches as several chunks in flight: ms per 50-step call for E episodes split into `lanes` equal chunks.
python tools/small_batch_lanes.py [precision]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
prec = sys.argv[1] if len(sys.argv) > 1 else "f16mx"
A, K, T = 5, 20, 12
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 23), joint=True, step=50)
for E in (2, 4, 8, 16, 32, 48):
    g = torch.Generator().manual_seed(3)
    ctx = torch.randn([E, A, 256], generator=g).cuda()
    x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
    ref = None
    row = []
    for lanes in (1, 2, 3, 4):
        if lanes > E:
            continue
        chunk = (E + lanes - 1) // lanes if lanes > 1 else 0
        eng.set_tuning("lanes", lanes)
        eng.set_chunk_episodes(chunk)
        v = eng.denoise(x_T, ctx, None, precision=prec, want_pos=False)[0]
        eng.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            v = eng.denoise(x_T, ctx, None, precision=prec, want_pos=False)[0]
        eng.synchronize(); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        if ref is None:
            ref = v.clone()
        row.append(f"{lanes} x {chunk or E}: {ms:.2f} ms ({E * A * K / ms:.1f} k traj/s){'' if torch.equal(v, ref) else ' DIFFERENT'}")
    print(f"[{prec}] E={E}: " + " | ".join(row), flush=True)
