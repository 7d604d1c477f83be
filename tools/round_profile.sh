#!/bin/bash
# All measurements that profiles/ holds for a round, in one GPU-box call.  Output: gpurun_out/final/
#   tools/round_profile.sh            (about 6-8 minutes)
#   JMID_PREC=f16x3 SKIP_BENCH=1 tools/round_profile.sh   -> kernel stats and PMC traffic of the other mode
export TMPDIR=/tmp
export JMID_PREC=${JMID_PREC:-f16x2}
P=$JMID_PREC
O=gpurun_out/final
mkdir -p $O
if [ -z "$SKIP_BENCH" ]; then
timeout 400 python bench.py --precision f16x2 --cpu-episodes 3 > $O/bench_cfg3_f16x2.log 2>&1
timeout 400 python bench.py --precision f16x3 --cpu-episodes 3 > $O/bench_cfg3_f16x3.log 2>&1
timeout 300 python bench.py --precision f32 --cpu-episodes 0 --steps 2 > $O/bench_cfg3_f32.log 2>&1
timeout 300 python bench.py --workload cfg2 --cpu-episodes 0 --steps 20 --warmup 3 > $O/bench_cfg2.log 2>&1
timeout 300 python bench.py --workload cfg4 --cpu-episodes 0 --steps 3 > $O/bench_cfg4.log 2>&1
timeout 400 python bench.py --workload cfg5 --cpu-episodes 0 --steps 2 > $O/bench_cfg5_1gpu.log 2>&1
timeout 300 python bench.py --net imid --cpu-episodes 0 --steps 2 > $O/bench_cfg3_imid.log 2>&1
fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --precision $P --lanes 1 --steps 1 --warmup 0 --cpu-episodes 0 --episodes-per-gpu 51 > $O/prof_bench_$P.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/${P}_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_attn_$c -- python tools/attn_only.py > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_gemm_$c -- python tools/gemm_only.py 0 > /dev/null 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_call_$c -- python tools/step_only.py 51 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, json, collections
out = {}
for tag in ("attn", "gemm", "call"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(list)
        for f in glob.glob(f"gpurun_out/final/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == c:
                    acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[f"{tag}:{c}:{k}"] = {"launches": len(v), "avg": sum(v) / len(v)}
        if tag == "call":
            out[f"call_total:{c}"] = sum(sum(v) for v in acc.values())
import os
json.dump(out, open(f"gpurun_out/final/pmc_raw_{os.environ['JMID_PREC']}.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith('call:')}, indent=1)[:4000])
PY
for f in $O/bench_*.log; do echo "== $f"; tail -1 $f | cut -c1-260; done
rm -rf $O/prof $O/pmc_attn_* $O/pmc_gemm_* $O/pmc_call_*
