import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 23), joint=True, step=4)
rng = np.random.default_rng(0)
for (N, K) in [(256, 512), (512, 512), (128, 256), (1536, 512)]:
    A = rng.standard_normal((40000, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    for prec, variants in (("f32", [0]), ("f16x3", [1, 2, 3])):
        for v in variants:
            eng.set_tuning("gemm_h_variant", v)
            outs = {}
            for M in (320, 384, 36000 + 320, 36000 + 384):
                outs[M] = eng.dbg_gemm(A[:M], W, b, precision=prec)
            d1 = np.abs(outs[320][256:320] - outs[384][256:320]).max()
            d2 = np.abs(outs[36320][36256:36320] - outs[36384][36256:36320]).max()
            d3 = np.abs(outs[320][:256] - outs[36384][:256]).max()
            print(f"N={N} K={K} {prec} v{v}: partial-vs-full tile small={d1:.3e} large={d2:.3e}  smallM-vs-largeM rows0-255={d3:.3e}", flush=True)
