"""Diagnostics for gemm_ln2_mx_kernel: with W = I, X = 0, bias = 0, gamma = 1, beta = 0 and N(0, 1) rows the output is ~A itself
(LayerNorm of a standard-normal row is nearly the identity): which input column does every output column carry?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 1), joint=True)
M, K = 128, 512
rng = np.random.default_rng(0)
A = rng.standard_normal((M, K)).astype(np.float32)
W = np.eye(512, K, dtype=np.float32)
z, o = np.zeros(512, np.float32), np.ones(512, np.float32)
X = np.zeros((M, 512), np.float32)
for fused in (False, True):
    out = eng.dbg_gemm_ln_mx(A, W, z, o, z, X, fused=fused)
    v = A.astype(np.float64)
    ref = (v - v.mean(1, keepdims=True)) / np.sqrt(v.var(1, keepdims=True) + 1e-5)
    print("fused" if fused else "pair ", "max err vs ref", np.abs(out - ref).max(), "finite", np.isfinite(out).all())
    if np.abs(out - ref).max() > 1e-2:
        c = (out - out.mean(0)).T @ (ref - ref.mean(0)) / M          # [out col, in col]
        src = c.argmax(1)
        print(" source column of output columns 0..63:", src[:64].tolist())
        print(" match quality:", np.round(c.max(1)[:16], 2).tolist())
        r = (out - out.mean(1, keepdims=True)) @ (ref - ref.mean(1, keepdims=True)).T / 512
        print(" source row of output rows 0..31:", r.argmax(1)[:32].tolist(), np.round(r.max(1)[:8], 2).tolist())
        print(" row 0, cols 0..15: out", np.round(out[0, :16], 3).tolist())
        print("                    ref", np.round(ref[0, :16], 3).tolist())
eng.close()
