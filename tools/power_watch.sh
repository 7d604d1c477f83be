#!/bin/bash
# sample rocm-smi power / clocks while the bench runs (GPU box)
rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -v "^$" | head -30
(timeout 120 python bench.py --steps 30 --warmup 1 --cpu-episodes 0 --no-profile > gpurun_out/pw_bench.log 2>&1) &
BP=$!
sleep 12
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|junction\|mclk" | head -8; echo ---; sleep 1.5; done
wait $BP
tail -1 gpurun_out/pw_bench.log | cut -c1-200
