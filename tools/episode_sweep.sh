#!/bin/bash
# throughput vs episodes per GPU (weak-scaling unit of the multi-GPU sweep), F16X3 mode
for e in 1 4 8 16 32 51 102 256 512; do
  timeout 300 python bench.py --cpu-episodes 0 --episodes-per-gpu $e --steps 2 --warmup 1 --no-profile 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('episodes/GPU', $e, 'traj/s', d['value'], 'ms/step', d['ms_per_step'])"
done
