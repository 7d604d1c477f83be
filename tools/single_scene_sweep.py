"""Single-scene (BASELINE configs[1]) latency under tuning knobs:  python tools/single_scene_sweep.py key=v1,v2,... [precision]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims

key, vals = sys.argv[1].split("=")
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
E = int(sys.argv[3]) if len(sys.argv) > 3 else 1
# the shape: BASELINE cfg2 by default; SS_SHAPE=A,K,T,steps for another (the reference's shipped point: SS_SHAPE=3,100,8,2)
A, K, T, STEP = (int(v) for v in os.environ.get("SS_SHAPE", "5,20,12,50").split(","))
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 0), joint=True, step=STEP)
g = torch.Generator().manual_seed(3)
ctx = torch.randn([E, A, 256], generator=g).cuda()
x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
p0 = torch.randn([E, A, 2], generator=g).cuda()
ref = None
for v in vals.split(","):
    eng.set_tuning(key, int(v))
    for _ in range(5):
        out = eng.denoise(x_T, ctx, p0, precision=prec, want_vel=False)[1]
    eng.synchronize()
    t = time.perf_counter()
    for _ in range(20 if STEP > 10 else 200):
        out = eng.denoise(x_T, ctx, p0, precision=prec, want_vel=False)[1]
    eng.synchronize()
    dt = (time.perf_counter() - t) / (20 if STEP > 10 else 200)
    if ref is None:
        ref = out.clone()
    print(f"{prec} E={E} {key}={v}: {1e3 * dt:.3f} ms/call  {E * A * K / dt:.0f} traj/s  max|d| vs first {float((out - ref).abs().max()):.2e}", flush=True)
