import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
A, K, T = 5, 20, 12
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 23), joint=True, step=50)
for prec in ("f16mx",):
    for E, chunk in ((250, 25), (260, 26), (240, 24), (255, 51), (250, 25), (260, 26), (230, 23), (270, 27)):
        g = torch.Generator().manual_seed(3)
        ctx = torch.randn([E, A, 256], generator=g).cuda()
        x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
        eng.set_chunk_episodes(chunk)
        v = eng.denoise(x_T, ctx, None, precision=prec, want_pos=False)[0]
        eng.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            v = eng.denoise(x_T, ctx, None, precision=prec, want_pos=False)[0]
        eng.synchronize(); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print(f"[{prec}] E={E} chunk={chunk}: {ms:.1f} ms = {ms / E:.4f} ms / episode = {E * A * K / ms:.2f} k traj/s", flush=True)
