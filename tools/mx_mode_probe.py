"""JMID_PREC_F16MX against the other modes: (1) one GEMM vs fp64, (2) a 51-episode 50-step call: mean / worst ADE vs the
exact-fp32 mode of the same library and time per call.   python tools/mx_mode_probe.py [episodes] [knob=value ...]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims

E = int(sys.argv[1]) if len(sys.argv) > 1 else 51
A, K, T = 5, 20, 12
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 23), joint=True, step=50)
for kv in sys.argv[2:]:                      # tuning knobs: key=value
    k, v = kv.split("=")
    eng.set_tuning(k, int(v))

rng = np.random.default_rng(0)
M, N, Kd = 16384, 1024, 512
Am = rng.standard_normal((M, Kd)).astype(np.float32)
Wm = (rng.standard_normal((N, Kd)) / np.sqrt(Kd)).astype(np.float32)
bias = rng.standard_normal(N).astype(np.float32)
ref = Am[:512].astype(np.float64) @ Wm.astype(np.float64).T + bias
for prec in ("f16x3", "f16x2", "f16mx"):
    out = eng.dbg_gemm(Am, Wm, bias, precision=prec)[:512]
    err = np.sqrt(((out - ref) ** 2).mean() / (ref ** 2).mean())
    print(f"GEMM {M}x{N}x{Kd} [{prec}]: rms error / rms value = {err:.3e} (2^{np.log2(err):.1f}), max abs {np.abs(out - ref).max():.3e}", flush=True)

g = torch.Generator().manual_seed(3)
ctx = torch.randn([E, A, 256], generator=g).cuda()
x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
exact = eng.denoise(x_T, ctx, None, precision="f32", want_pos=False)[0].clone()
for prec in ("f16x3", "f16x2", "f16mx"):
    v = eng.denoise(x_T, ctx, None, precision=prec, want_pos=False)[0]
    eng.synchronize()
    d = (v - exact).norm(dim=-1)
    per_ep = d.reshape(E, -1).mean(dim=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.denoise(x_T, ctx, None, precision=prec, want_pos=False)
    eng.synchronize()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{E} episodes [{prec}]: mean ADE vs f32 mode {float(d.mean()):.3e}, worst episode {float(per_ep.max()):.3e}, worst point "
          f"{float(d.max()):.3e}; {ms:.2f} ms / call = {E * A * K / ms * 1e3:.0f} traj/s", flush=True)
