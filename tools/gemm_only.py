import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=32), 0), joint=True)
rng = np.random.default_rng(0)
M, N, K = 61440, 1536, 512
A = rng.standard_normal((M, K)).astype(np.float32); W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
b = rng.standard_normal(N).astype(np.float32)
v = int(sys.argv[1]) if len(sys.argv) > 1 else 4
eng.set_tuning("gemm_h_variant", v)
for _ in range(3):
    eng.dbg_gemm(A, W, b, precision=os.environ.get("JMID_PREC", "f16x2"))
