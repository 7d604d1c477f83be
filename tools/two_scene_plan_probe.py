"""Two cfg2 scenes per call in the modes that do not take the one-launch GEMM + LayerNorm (f16x2, f16x3): two halves side by side
(the default plan) against ONE chunk, with the split-KV factor of the halves (6) and the one picked for the 80-block launch (3).
Run on the GPU box with the diagnostics flavour:  JMID_LIB=.../libjmid_hip_diag.so python tools/two_scene_plan_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims

E, A, K, T = int(os.environ.get("PROBE_E", "2")), 5, 20, 12
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 0), joint=True, step=50)
g = torch.Generator().manual_seed(3)
ctx = torch.randn([E, A, 256], generator=g).cuda()
x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
for prec in sys.argv[1:] or ["f16x3", "f16x2", "f16mx"]:
    for lanes, ns in ((2, 0), (1, 0), (1, 3), (2, 3), (2, 0), (1, 3)):
        eng.set_tuning("lanes", lanes)
        eng.set_tuning("attn_nsplit", ns)
        for _ in range(4):
            eng.denoise(x_T, ctx, precision=prec, want_vel=False)
        eng.synchronize()
        t = time.perf_counter()
        for _ in range(15):
            eng.denoise(x_T, ctx, precision=prec, want_vel=False)
        eng.synchronize()
        print(f"{prec} E={E} lanes={lanes} attn_nsplit={ns}: {1e3 * (time.perf_counter() - t) / 15:.3f} ms/call", flush=True)
    eng.set_tuning("lanes", 2)
    eng.set_tuning("attn_nsplit", 0)
