"""Single-scene call: wall time and host CPU time per call with the eager launch chain vs the captured hipGraph loop.
   python tools/graph_latency.py [precision] [episodes]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 1
A, K, T = 5, 20, 12
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 0), joint=True, step=50)
g = torch.Generator().manual_seed(3)
ctx = torch.randn([E, A, 256], generator=g).cuda()
x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
p0 = torch.randn([E, A, 2], generator=g).cuda()
for mode, name in ((2, "eager launches"), (1, "captured graph")):
    eng.set_tuning("graph", mode)
    for _ in range(5):
        eng.denoise(x_T, ctx, p0, precision=prec, want_vel=False)
    eng.synchronize()
    n = 30
    t, c = time.perf_counter(), time.process_time()
    for _ in range(n):
        eng.denoise(x_T, ctx, p0, precision=prec, want_vel=False)
    eng.synchronize()
    dt, dc = (time.perf_counter() - t) / n, (time.process_time() - c) / n
    print(f"{prec} E={E} {name}: wall {1e3 * dt:.3f} ms/call, host CPU {1e3 * dc:.3f} ms/call, {E * A * K / dt:.0f} traj/s, "
          f"graph replays so far {eng.graph_replays()}", flush=True)
