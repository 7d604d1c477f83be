import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
prec = sys.argv[1] if len(sys.argv) > 1 else "f16mx"
E, A, K, T = 1, 5, 20, 12
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 0), joint=True, step=50)
g = torch.Generator().manual_seed(3)
ctx = torch.randn([E, A, 256], generator=g).cuda()
x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
p0 = torch.randn([E, A, 2], generator=g).cuda()
for graph in (0, 1):
    eng.set_tuning("graph", graph)
    for _ in range(5):
        eng.denoise(x_T, ctx, p0, precision=prec, want_vel=False)
    eng.synchronize()
    enq, tot = [], []
    for _ in range(20):
        t0 = time.perf_counter()
        eng.denoise(x_T, ctx, p0, precision=prec, want_vel=False)
        t1 = time.perf_counter()
        eng.synchronize()
        t2 = time.perf_counter()
        enq.append(t1 - t0); tot.append(t2 - t0)
    print(f"{prec} graph={graph}: enqueue {1e3*sum(enq)/len(enq):.3f} ms, total {1e3*sum(tot)/len(tot):.3f} ms, replays {eng.graph_replays()}")
