"""f16mx GEMM against f16x2 / fp64 over the tile shapes (dbg entry): python tools/mx_shapes.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims

eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=32), 1), joint=True, step=2)
rng = np.random.default_rng(0)
for variant in (0, 3, 4, 5, 6):
    eng.set_tuning("gemm_h_variant", variant)
    for (M, N, K) in ((1200, 512, 512), (1200, 1536, 512), (15600, 1536, 512), (15600, 1024, 512), (15600, 512, 1024), (9000, 256, 512), (70000, 1024, 512), (333, 128, 256)):
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        ref = A.astype(np.float64) @ W.astype(np.float64).T + b
        o2 = eng.dbg_gemm(A, W, b, precision="f16x2")
        ox = eng.dbg_gemm(A, W, b, precision="f16mx")
        e2 = np.sqrt(((o2 - ref) ** 2).mean() / (ref ** 2).mean())
        ex = np.sqrt(((ox - ref) ** 2).mean() / (ref ** 2).mean())
        bad = int((~np.isfinite(ox)).sum())
        print(f"variant {variant} M {M} N {N} K {K}: f16x2 {e2:.2e}  f16mx {ex:.2e}  max|mx - x2| {np.abs(ox - o2).max():.2e}  nonfinite {bad}", flush=True)
