"""Parity and time of the arithmetic modes on the cfg3 shape (A=5, K=20, T=12, 50 steps), against the exact-fp32 mode
   of the same library:   python tools/precision_modes.py [f16x3 f16x2 ...]      (PM_EPISODES, PM_NET=jmid|imid)
   PM_QK_SCALE=s multiplies the Q and K rows of every in_proj weight and bias by s (softmax logits x s^2): peaked
   attention as a trained network may have, where rounded logits would hurt."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims

modes = sys.argv[1:] or ["f16x3", "f16x2"]
E, A, K, T = int(os.environ.get("PM_EPISODES", "51")), 5, 20, 12
joint = os.environ.get("PM_NET", "jmid") == "jmid"
g = torch.Generator().manual_seed(5)
ctx = torch.randn([E, A, 256], generator=g).cuda()
x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
p0 = (4 * torch.randn([E, A, 2], generator=g)).cuda()
qk = float(os.environ.get("PM_QK_SCALE", "1"))
for seed in (0, 1, 2):
    w = JMIDWeights.from_seed(NetDims(ctx_dim=256), seed)
    if qk != 1.0:
        for k, t in w.tensors.items():
            if "in_proj" in k:
                t[: 2 * t.shape[0] // 3] *= qk
    eng = JmidEngine(w, joint=joint, step=50)
    ref = eng.denoise(x_T, ctx, p0, precision="f32")[1].cpu().numpy()
    for mode in modes:
        eng.denoise(x_T, ctx, p0, precision=mode)
        eng.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            pos = eng.denoise(x_T, ctx, p0, precision=mode)[1]
            eng.synchronize()
            ts.append(time.perf_counter() - t0)
        err = np.linalg.norm(pos.cpu().numpy() - ref, axis=-1)
        print(f"qk x{qk:g} weights seed {seed} {mode:6s}: mean ADE {err.mean():.2e} m, worst episode {err.reshape(E, -1).mean(1).max():.2e}, "
              f"worst point {err.max():.2e} | {1e3 * min(ts):7.1f} ms per call, {E * A * K / min(ts):8.0f} traj/s", flush=True)
