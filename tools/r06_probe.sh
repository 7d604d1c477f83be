#!/bin/bash
# round 6: (1) the tests that failed in the first full run, (2) the de-phasing experiment of the F16MX GEMM + LayerNorm ("ln_stagger"), (3) cycle
# stamps of the attention kernel's key-tile loop with both softmax forms
O=gpurun_out/${R:-r06e}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_robustness.py tests/test_gpu_forecaster.py "tests/test_gpu_parity.py::test_packed_short_sequence_attention_matches_unpacked" -q -m gpu > $O/retest.log 2>&1; echo "rc $?" >> $O/retest.log
(for K in 512 1024; do echo "== K = $K: 128-row tiles, first-round workgroups of every second CU start n x 3.9 us late (variant 1000 + n)"; timeout 200 build/gemm_ln2_check 61200 $K 20 128,1001,1002,1003,1004,1006; done) 2>&1 | grep -v amdgpu.ids > $O/gemm_ln2_stagger.log
(TRACE=1 timeout 200 build/attn_check 51 1200 10 0; echo "== one workgroup per CU"; ONE_WG=1 TRACE=1 timeout 200 build/attn_check 51 1200 10 0) 2>&1 | grep -v amdgpu.ids | grep "traced\|ms per launch\|==" > $O/attn_trace.log
tail -8 $O/retest.log; cat $O/gemm_ln2_stagger.log $O/attn_trace.log
