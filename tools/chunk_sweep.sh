#!/bin/bash
# bench throughput vs episodes-per-pass (workgroup-count quantisation of the attention launch)
for c in 12 20 25 38 51; do
  python bench.py --precision f16x3 --cpu-episodes 0 --episodes-per-gpu 102 --steps 1 --warmup 1 --chunk $c 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels']
print('chunk', $c, 'traj/s', d['value'], 'attn', k['attention']['tflops'], 'qkv', k['gemm_qkv']['tflops'], 'ff1', k['gemm_ff1']['tflops'], 'ff2', k['gemm_ff2']['tflops'], 'frac', d['kernel_time_fraction_of_step'])"
done
