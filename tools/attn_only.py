import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 0), joint=True)
rng = np.random.default_rng(0)
nseq, S, d = 51, 1200, 512
qkv = rng.standard_normal((nseq * S, 3 * d)).astype(np.float32)
for _ in range(3):
    eng.dbg_attention(qkv, nseq, S, precision=sys.argv[1] if len(sys.argv) > 1 else os.environ.get("JMID_PREC", "f16x2"))
