"""Host enqueue time against completion time of a 50-step call: is a small batch bound by the host's launch rate?
python tools/enqueue_time.py [precision]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
prec = sys.argv[1] if len(sys.argv) > 1 else "f16mx"
A, K, T = 5, 20, 12
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 23), joint=True, step=50)
for E, lanes in ((1, 1), (2, 2), (4, 1), (4, 2), (4, 4), (8, 2), (8, 4), (16, 2), (51, 1)):
    g = torch.Generator().manual_seed(3)
    ctx = torch.randn([E, A, 256], generator=g).cuda()
    x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
    eng.set_tuning("lanes", lanes)
    eng.set_chunk_episodes((E + lanes - 1) // lanes if lanes > 1 else 0)
    eng.denoise(x_T, ctx, None, precision=prec, want_pos=False)
    eng.synchronize(); torch.cuda.synchronize()
    enq = tot = 0.0
    for _ in range(5):
        t0 = time.perf_counter()
        eng.denoise(x_T, ctx, None, precision=prec, want_pos=False)
        t1 = time.perf_counter()
        eng.synchronize(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq += t1 - t0; tot += t2 - t0
    print(f"[{prec}] E={E} lanes={lanes}: enqueue {enq / 5 * 1e3:.2f} ms, complete {tot / 5 * 1e3:.2f} ms", flush=True)
