#!/bin/bash
# The one-scene measurements of a round in one GPU-box call (the rest of profiles/ comes from tools/round_profile.sh): the default bench
# line, the cfg2 bench, rocprofv3 kernel stats of the cfg2 workload, the knob A/B of the launches folded into the one-launch GEMM +
# LayerNorm (gemm_small.hpp, OUT_LNX / lnx_combine), the MPC-sized episode counts.   Output: gpurun_out/$R/
export TMPDIR=/tmp
R=${R:-r05s}
O=gpurun_out/$R
mkdir -p $O
timeout 900 python bench.py --detail $O/bench_cfg3.json > $O/bench_cfg3.line.json 2> $O/bench_cfg3.err
timeout 300 python bench.py --no-pmc --workload cfg2 --cpu-episodes 0 --no-e2e --steps 20 --warmup 3 --detail $O/bench_cfg2.json > $O/bench_cfg2.line.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ss -- python bench.py --no-pmc --workload cfg2 --modes f16mx --steps 20 --warmup 3 --cpu-episodes 0 --no-e2e --no-profile > $O/prof_bench_cfg2.log 2>&1
cp "$(find $O/prof_ss -name "*kernel_stats.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2-)" $O/cfg2_f16mx_kernel_stats.csv
rm -rf $O/prof_ss
export JMID_LIB=safe-interactive-crowdnav_amd/csrc/libjmid_hip_diag.so
(python tools/single_scene_sweep.py small_lnx=2,0,1,2,0,1 f16mx; python tools/single_scene_sweep.py small_cmb=2,0,2,0 f16mx; python tools/single_scene_sweep.py small_lnx=2,0 f16x3) 2>/dev/null | grep ms > $O/one_scene_knobs.log
unset JMID_LIB
for e in 1 2 4 8; do
  timeout 300 python bench.py --no-pmc --cpu-episodes 0 --no-e2e --episodes-per-gpu $e --steps 2 --warmup 1 --no-profile 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('episodes/call', $e, {m: (v['value'], v['ms_per_step']) for m, v in d['modes'].items()})"
done > $O/episode_sweep.log
cat $O/bench_cfg3.line.json; cat $O/one_scene_knobs.log $O/episode_sweep.log; cut -c1-150 $O/cfg2_f16mx_kernel_stats.csv | head -12
