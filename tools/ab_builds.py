"""A/B of two (or more) builds of libjmid_hip.so inside one process (same box, same clocks):
     git stash / checkout the other source, hipcc ... -o build/libjmid_a.so, restore, then
     python tools/ab_builds.py f16x3 build/libjmid_a.so safe-interactive-crowdnav_amd/csrc/libjmid_hip.so
   Alternates the builds over full 50-step calls on one 51-episode chunk and prints ms per call of each."""
import os, sys, time
import ctypes as C
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from safe_interactive_crowdnav_amd import _lib
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims

mode, paths = sys.argv[1], sys.argv[2:]
E, A, K, T = int(os.environ.get("AB_EPISODES", "51")), 5, 20, 12
g = torch.Generator().manual_seed(5)
ctx = torch.randn([E, A, 256], generator=g).cuda()
x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
engs = []
for p in paths:
    engs.append(JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 0), joint=os.environ.get("AB_NET", "jmid") == "jmid", step=50,
                           lib_path=os.path.join(ROOT, p)))
res = [[] for _ in paths]
outs = [None] * len(paths)
for rep in range(int(os.environ.get("AB_REPS", "5"))):
    for i, eng in enumerate(engs):
        eng.synchronize()
        t0 = time.perf_counter()
        outs[i] = eng.denoise(x_T, ctx, None, precision=mode, want_pos=False)[0]
        eng.synchronize()
        if rep:
            res[i].append(time.perf_counter() - t0)
for i, p in enumerate(paths):
    ms = 1e3 * np.array(res[i])
    same = bool(torch.equal(outs[i], outs[0]))
    print(f"{mode} {p}: {ms.mean():8.2f} ms per call (min {ms.min():.2f})  {E * A * K / ms.mean() * 1e3:9.0f} traj/s  same bits as first: {same}", flush=True)
