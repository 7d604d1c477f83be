"""Two cfg2 scenes per call: the default plan (ONE chunk, the split-KV factor of that launch) against the plan before it (two halves side by
side, 6 key ranges), alternating on one box.  JMID_LIB=.../libjmid_hip_diag.so python tools/two_scene_ab.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
E, A, K, T = 2, 5, 20, 12
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 0), joint=True, step=50)
g = torch.Generator().manual_seed(3)
ctx = torch.randn([E, A, 256], generator=g).cuda(); x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
for prec in ("f16x2", "f16mx"):
    for chunk, ns in ((0, 0), (1, 6), (0, 0), (1, 6), (0, 0)):
        eng.set_chunk_episodes(chunk); eng.set_tuning("attn_nsplit", ns)
        for _ in range(4): eng.denoise(x_T, ctx, precision=prec, want_vel=False)
        eng.synchronize(); t = time.perf_counter()
        for _ in range(15): eng.denoise(x_T, ctx, precision=prec, want_vel=False)
        eng.synchronize()
        print(f"{prec} E=2 chunk={chunk} attn_nsplit={ns}: {1e3 * (time.perf_counter() - t) / 15:.3f} ms/call", flush=True)
