#!/bin/bash
export TMPDIR=/tmp
R=r02; O=gpurun_out/$R; mkdir -p $O
for m in f16mx f16x2 f16x3; do
  JMID_PREC=$m tools/pmc_call.sh > $O/pmc_$m.log 2>&1
  cp gpurun_out/pmc/pmc_call_$m.json $O/
  cp gpurun_out/pmc/pmc_call_$m.json profiles/${R}_pmc_call_$m.json
done
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
tail -2 $O/bench_cfg3.err
for m in f16mx f16x2 f16x3; do tail -3 $O/pmc_$m.log | head -1; done
