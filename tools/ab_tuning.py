"""A/B of a tuning key inside one process (same box, same clocks): full 50-step denoise on one 51-episode chunk.
   python tools/ab_tuning.py ln_fuse 2 1        -> alternates value 2 and value 1, prints ms per call of each"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims

key, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
E, N, K, H = int(os.environ.get("AB_EPISODES", "51")), 5, 20, 12
dev = torch.device("cuda", 0)
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 0), joint=os.environ.get("AB_NET", "jmid") == "jmid", step=50)
ctx = torch.randn([E, N, 256], generator=torch.Generator().manual_seed(1)).to(dev)
x_T = torch.randn([E, K * N, H, 2], generator=torch.Generator().manual_seed(0)).to(dev)
if os.environ.get("AB_CHUNK"):
    eng.set_chunk_episodes(int(os.environ["AB_CHUNK"]))
if os.environ.get("AB_LANES"):
    eng.set_tuning("lanes", int(os.environ["AB_LANES"]))
for kv in os.environ.get("AB_PRESET", "").split(","):   # AB_PRESET="key=value,key=value": fixed for both arms
    if kv:
        eng.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
res = {v: [] for v in vals}
for rep in range(4):
    for v in vals:
        eng.set_tuning(key, v)
        eng.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.denoise(x_T, ctx, None, precision=os.environ.get("AB_PREC", "f16x2"), want_pos=False)
        eng.synchronize()
        if rep:
            res[v].append(time.perf_counter() - t0)
for v in vals:
    ms = 1e3 * np.array(res[v])
    print(f"{key}={v}: {ms.mean():8.2f} ms per call (min {ms.min():.2f})  {E*N*K/ms.mean()*1e3:9.0f} traj/s", flush=True)
