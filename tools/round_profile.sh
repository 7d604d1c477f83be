#!/bin/bash
# All measurements profiles/ holds for a round, in one GPU-box call (about 15 minutes).  Output: gpurun_out/$R/
#   R=r02 tools/round_profile.sh ; then python tools/update_profiles.py r02
export TMPDIR=/tmp
R=${R:-r06}
O=gpurun_out/$R
mkdir -p $O
# PMC first: HBM bytes and MFMA busy per production kernel over one whole call, per mode; the summaries go into profiles/ of
# this copy right away, so that the bench lines below quote THEM (file name + git blob hash of exactly these bytes)
for m in f16mx f16x2 f16x3; do
  JMID_PREC=$m tools/pmc_call.sh > $O/pmc_$m.log 2>&1 || { echo "PMC pass failed ($m)"; cat $O/pmc_$m.log; exit 1; }
  cp gpurun_out/pmc/pmc_call_$m.json $O/
  cp gpurun_out/pmc/pmc_call_$m.json profiles/${R}_pmc_call_$m.json
done
# default bench (BASELINE configs[2]), all split modes measured identically, parity sample over all chunks
timeout 600 python bench.py --steps 10 --warmup 3 --detail $O/bench_cfg3.json > $O/bench_cfg3.line.json 2> $O/bench_cfg3.err
timeout 300 python bench.py --no-pmc --precision f32 --modes f32 --cpu-episodes 0 --no-e2e --steps 2 --detail $O/bench_cfg3_f32.json > $O/bench_cfg3_f32.line.json 2>/dev/null
timeout 300 python bench.py --no-pmc --workload cfg2 --cpu-episodes 0 --no-e2e --steps 20 --warmup 3 --detail $O/bench_cfg2.json > $O/bench_cfg2.line.json 2>/dev/null
timeout 300 python bench.py --no-pmc --workload cfg4 --cpu-episodes 0 --no-e2e --steps 3 --detail $O/bench_cfg4.json > $O/bench_cfg4.line.json 2>/dev/null
timeout 400 python bench.py --no-pmc --workload cfg5 --cpu-episodes 0 --no-e2e --steps 2 --detail $O/bench_cfg5_1gpu.json > $O/bench_cfg5_1gpu.line.json 2>/dev/null
timeout 300 python bench.py --no-pmc --net imid --cpu-episodes 0 --no-e2e --steps 2 --detail $O/bench_cfg3_imid.json > $O/bench_cfg3_imid.line.json 2>/dev/null
timeout 300 python bench.py --no-pmc --scenes orca --cpu-episodes 4 --no-e2e --steps 2 --detail $O/bench_cfg3_orca.json > $O/bench_cfg3_orca.line.json 2>/dev/null
# episodes per call sweep (weak-scaling unit), both modes
for e in 1 2 4 8 16 32 52 104 256 512; do
  timeout 300 python bench.py --no-pmc --cpu-episodes 0 --no-e2e --episodes-per-gpu $e --steps 2 --warmup 1 --no-profile 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('episodes/call', $e, {m: (v['value'], v['ms_per_step']) for m, v in d['modes'].items()})"
done > $O/episode_sweep.log
# rocprofv3 kernel stats of one step on a 51-episode chunk, per mode
for m in f16mx f16x2 f16x3; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -- python bench.py --no-pmc --precision $m --modes $m --lanes 1 --steps 1 --warmup 0 --cpu-episodes 0 --no-e2e --episodes-per-gpu 51 > $O/prof_bench_$m.log 2>&1
  cp "$(find $O/prof_$m -name "*kernel_stats.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2-)" $O/${m}_kernel_stats.csv
  rm -rf $O/prof_$m
done
# single-scene kernel stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ss -- python bench.py --no-pmc --workload cfg2 --modes f16mx --steps 20 --warmup 3 --cpu-episodes 0 --no-e2e --no-profile > $O/prof_bench_cfg2.log 2>&1
cp "$(find $O/prof_ss -name "*kernel_stats.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2-)" $O/cfg2_f16mx_kernel_stats.csv
rm -rf $O/prof_ss
# round 4: the shipped operating point end to end (wall split + kernel stats), the one-scene launch traces, the 2-rank strong-scaling line
for m in f16mx f16x3; do JMID_PREC=$m tools/shipped_profile.sh; done > $O/shipped_profile.log 2>&1
JMID_PREC=f16mx tools/small_pmc.sh 1 > /dev/null 2>&1; cp gpurun_out/small_pmc_f16mx_E1.txt $O/small_pmc_f16mx_E1.txt
for t in small_gemm_trace_0 attn_small_trace; do if [ -x build/$t ]; then ./build/$t; fi; done > $O/small_launch_traces.log 2>&1
timeout 600 python bench.py --gpus 2 --dist-backend gloo --device 0 --total-episodes 512 --steps 2 --warmup 1 --cpu-episodes 0 --no-profile --detail $O/bench_strong_2ranks_1gpu.json 2>/dev/null | tail -1 > $O/bench_strong_2ranks_1gpu.line.json
# reproducibility soak of the default path + the documented multi-lane disturbance
# reproducibility soak: one chunk in flight, and 2 / 3 / 4 chunks in flight against the one-chunk reference (bitwise)
python tools/rerun_soak.py f16mx 20 256 1 > $O/soak.log 2>&1
python tools/rerun_soak.py f16mx 20 256 2 >> $O/soak.log 2>&1
python tools/rerun_soak.py f16mx 10 256 3 >> $O/soak.log 2>&1
python tools/rerun_soak.py f16mx 12 64 2 32 >> $O/soak.log 2>&1
python tools/rerun_soak.py f16x2 20 256 1 >> $O/soak.log 2>&1
python tools/rerun_soak.py f16x3 20 256 1 >> $O/soak.log 2>&1
python tools/rerun_soak.py f16x2 20 256 2 >> $O/soak.log 2>&1
python tools/rerun_soak.py f16x3 20 256 2 >> $O/soak.log 2>&1
python tools/rerun_soak.py f16x2 10 256 3 >> $O/soak.log 2>&1
python tools/rerun_soak.py f16x3 10 256 4 >> $O/soak.log 2>&1
# the instruction-level reproduction of what used to disturb them (needs build/concurrency_probe8, tools/concurrency_probe8.hip)
if [ -x build/concurrency_probe8 ]; then for c in 3 1 0; do timeout 300 ./build/concurrency_probe8 300 $c; done > $O/packed_fp32_probe.log 2>&1; fi
# the second co-residency finding: the scaled fp8 MFMA pair next to out_ddim_kernel (needs build/concurrency_probe9 and, for the
# "before" half, build/concurrency_probe9_scaled: the same probe built with -DPROBE_SCALED_MFMA)
if [ -x build/concurrency_probe9 ]; then (timeout 300 ./build/concurrency_probe9 40 | grep -v "hi plane\|tiles with"; if [ -x build/concurrency_probe9_scaled ]; then echo "--- the same with the SCALED instruction pair (v_mfma_ld_scale_b32 + v_mfma_scale_f32_32x32x64_f8f6f4)"; timeout 300 ./build/concurrency_probe9_scaled 40 | grep -v "hi plane\|tiles with"; fi) > $O/scaled_mfma_probe.log 2>&1; fi
# round 5: the 8-wave ping-pong attention experiment against the shipped kernel (bitwise + time per launch, three operand sets; cycle stamps)
# where each arithmetic mode leaves the gate, incl. the two extreme cells (no JMID_ERANGE anywhere)
timeout 900 python tools/robustness_sweep.py --out $O/robustness.json > $O/robustness.log 2>&1
# round 6: what the waves of the big kernels spend their cycles on (SQ counters, one per pass), after the softmax change
JMID_PREC=f16mx tools/sq_counters.sh > $O/sq_counters.log 2>&1; cp gpurun_out/sq/sq_counters_f16mx.json $O/sq_counters_f16mx.json 2>/dev/null
# lanes 1 / 2 / 3 on the default batch
python tools/single_scene_sweep.py lanes=1,2,3 f16mx 256 2>/dev/null | grep ms > $O/lanes.log
python tools/single_scene_sweep.py lanes=1,2,3 f16x2 256 2>/dev/null | grep ms >> $O/lanes.log
python tools/single_scene_sweep.py lanes=1,2,3 f16x3 256 2>/dev/null | grep ms >> $O/lanes.log
for f in $O/bench_*.json; do echo "== $f"; python -c "
import json,sys
d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d.get('single_scene',{}).get('ms_per_call'), {m:(v['value'], v.get('parity',{}).get('mean_ADE_vs_oracle_m')) for m,v in d['modes'].items()})"; done
cat $O/episode_sweep.log; grep -v amdgpu $O/soak.log | tail -12
