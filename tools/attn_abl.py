# Timing ablations of the kernel (results are WRONG by construction).  NEEDS A BUILD WITH -DJMID_ABLATIONS - the production
# library has no ablation knobs:
#   (cd safe-interactive-crowdnav_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value \
#        -ffp-contract=off -DJMID_ABLATIONS -o ../../build/libjmid_abl.so jmid_abi.hip jmid_weights.hip jmid_planner.hip jmid_profile.hip jmid_diag.hip)
#   JMID_LIB=build/libjmid_abl.so python tools/attn_abl.py
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
from tools.microbench import time_class
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 0), joint=True)
rng = np.random.default_rng(0)
nseq, S, d = 51, 1200, 512
qkv = rng.standard_normal((nseq * S, 3 * d)).astype(np.float32)
names = {0: "full", 1: "no DMA (reuse tile 0)", 2: "no softmax VALU", 4: "no PV MFMA", 8: "no S MFMA", 3: "no DMA, no softmax",
         6: "no softmax, no PV", 12: "no MFMA at all", 14: "only DMA + barriers", 15: "barriers only", 13: "softmax only (no DMA/MFMA)"}
for abl, nm in names.items():
    eng.set_tuning("attn_abl", abl)
    ms = time_class(eng, "attention", lambda: eng.dbg_attention(qkv, nseq, S, precision="f16x3"), 3)
    print(f"abl={abl:2d} {nm:32s} {ms*1e3:8.1f} us", flush=True)
