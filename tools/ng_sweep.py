import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
from tools.microbench import time_class
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=32), 0), joint=True)
rng = np.random.default_rng(0)
for (M, N, K) in [(61440, 1536, 512), (61440, 1024, 512), (61440, 512, 1024)]:
    A = rng.standard_normal((M, K)).astype(np.float32); W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    eng.set_tuning("gemm_h_variant", 4)
    line = f"M={M} N={N} K={K}:"
    for ng in (1, 2, 3, 4, 6, 12):
        if (N // 128) % ng: continue
        eng.set_tuning("gemm_ng", ng)
        ms = time_class(eng, "gemm_qkv", lambda: eng.dbg_gemm(A, W, b, precision="f16x3"), 3)
        line += f" NG={ng}: {ms*1e3:6.1f}us {2.0*M*N*K/ms/1e9:6.1f}TF |"
    print(line, flush=True)
