# Timing ablations of the kernel (results are WRONG by construction).  NEEDS A BUILD WITH -DJMID_ABLATIONS - the production
# library has no ablation knobs:
#   (cd safe-interactive-crowdnav_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value \
#        -ffp-contract=off -DJMID_ABLATIONS -o ../../build/libjmid_abl.so jmid_abi.hip jmid_weights.hip jmid_planner.hip jmid_profile.hip jmid_diag.hip)
#   JMID_LIB=build/libjmid_abl.so python tools/gemm_abl.py
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
from tools.microbench import time_class
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=32), 0), joint=True)
rng = np.random.default_rng(0)
M, N, K = 61440, 1536, 512
A = rng.standard_normal((M, K)).astype(np.float32); W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
b = rng.standard_normal(N).astype(np.float32)
eng.set_tuning("gemm_h_variant", 4)
for abl, nm in {0: "full", 1: "DMA + barriers + epilogue only", 2: "DMA + fragment reads, no MFMA", 4: "no DMA (stale LDS)", 5: "barriers + epilogue only"}.items():
    eng.set_tuning("gemm_abl", abl)
    ms = time_class(eng, "gemm_qkv", lambda: eng.dbg_gemm(A, W, b, precision="f16x3"), 3)
    print(f"abl={abl} {nm:36s} {ms*1e3:8.1f} us", flush=True)
