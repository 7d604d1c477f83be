"""MPC-sized calls (2 ... 8 episodes): chunk size x chunks in flight x small-launch kernels with several chunks in flight.
    JMID_LIB=.../libjmid_hip_diag.so python tools/mpc_lanes_sweep.py [f16mx]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
prec = sys.argv[1] if len(sys.argv) > 1 else "f16mx"
A, K, T = 5, 20, 12
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 0), joint=True, step=50)
for E in (2, 4, 8):
    g = torch.Generator().manual_seed(3)
    ctx = torch.randn([E, A, 256], generator=g).cuda()
    x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
    p0 = torch.randn([E, A, 2], generator=g).cuda()
    for chunk, lanes, sl in [(0, 2, 0), (1, 2, 0), (1, 2, 1), (1, 3, 0), (1, 3, 1), (1, 4, 0), (1, 4, 1), (2, 4, 0), (2, 4, 1), (2, 3, 0)]:
        if chunk and chunk * lanes > E and not (chunk == 1 and lanes <= E):
            continue
        eng.set_chunk_episodes(chunk)
        eng.set_tuning("lanes", lanes)
        eng.set_tuning("small_lanes", sl)
        for _ in range(4):
            eng.denoise(x_T, ctx, p0, precision=prec, want_vel=False)
        eng.synchronize()
        t = time.perf_counter()
        for _ in range(15):
            eng.denoise(x_T, ctx, p0, precision=prec, want_vel=False)
        eng.synchronize()
        dt = (time.perf_counter() - t) / 15
        print(f"{prec} E={E} chunk={chunk or 'auto'} lanes={lanes} small_lanes={sl}: {1e3 * dt:.3f} ms/call  {E * A * K / dt:.0f} traj/s", flush=True)
