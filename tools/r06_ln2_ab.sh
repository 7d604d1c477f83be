#!/bin/bash
O=gpurun_out/r06g; mkdir -p $O
(for K in 512 1024; do for rep in 1 2; do for v in r05 old new; do echo "== K=$K $v (r05: two LDS passes, squared deviations from the row mean; old: block-wise, ds_bpermute exchange; new: block-wise, v_permlane32_swap + fused multiply-adds)"; timeout 100 build/gemm_ln2_check_$v 61200 $K 20 128 | grep "per launch"; done; done; done) 2>&1 | grep -v amdgpu.ids > $O/gemm_ln2_ab.log
cat $O/gemm_ln2_ab.log
