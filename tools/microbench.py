#!/usr/bin/env python3
"""Kernel micro-benchmarks through the diagnostics entry points (GPU box).  Times come from the library's own
HIP-event profiler (kernel only, host copies excluded)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine  # noqa: E402
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims  # noqa: E402


def time_class(eng, cls, fn, iters=5):
    fn()
    eng.profile_enable([cls])
    eng.profile_reset()
    for _ in range(iters):
        fn()
    n, ms = eng.profile_get()[cls]
    eng.profile_disable()
    return ms / max(n, 1)


def main():
    eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 0), joint=True)
    rng = np.random.default_rng(0)
    print("== GEMM  C[M,N] = A[M,K] W[N,K]^T ==")
    for (M, N, K) in [(24000, 1536, 512), (24000, 512, 512), (24000, 1024, 512), (24000, 512, 1024), (1200, 1536, 512)]:
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        fl = 2.0 * M * N * K
        ms = time_class(eng, "gemm_qkv", lambda: eng.dbg_gemm(A, W, b, precision="f32"), 3)
        line = f"M={M} N={N} K={K}: f32 {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF |"
        for v in (1, 5, 4):
            eng.set_tuning("gemm_h_variant", v)
            ms = time_class(eng, "gemm_qkv", lambda: eng.dbg_gemm(A, W, b, precision="f16x3"), 3)
            line += f" f16x3[v{v}] {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF |"
        eng.set_tuning("gemm_h_variant", 0)
        print(line, flush=True)
    print("== attention (4 heads x 128) ==")
    d = 512
    for (nseq, S) in [(20, 1200), (1, 1200), (4, 19200 // 4)]:
        qkv = rng.standard_normal((nseq * S, 3 * d)).astype(np.float32)
        fl = 4.0 * nseq * S * S * d
        line = f"nseq={nseq} S={S}:"
        for prec in ("f32", "f16x3"):
            ms = time_class(eng, "attention", lambda: eng.dbg_attention(qkv, nseq, S, precision=prec), 3)
            line += f" {prec} {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF |"
        print(line, flush=True)


if __name__ == "__main__":
    main()
