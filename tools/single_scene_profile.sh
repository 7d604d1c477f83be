#!/bin/bash
# kernel-time vs wall-time of the single-scene call (BASELINE configs[1]); run on the GPU box from the repo root
export TMPDIR=/tmp
mkdir -p gpurun_out/ss
timeout 300 python bench.py --workload cfg2 --steps 20 --warmup 3 --cpu-episodes 0 --no-profile 2>/dev/null | tail -1 > gpurun_out/ss/bench_plain.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ss/prof -- python bench.py --workload cfg2 --steps 20 --warmup 3 --cpu-episodes 0 --no-profile > gpurun_out/ss/bench_prof.log 2>&1
find gpurun_out/ss/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/ss/kernel_stats.csv \;
python - <<'PY'
import csv, json
j = json.loads(open('gpurun_out/ss/bench_plain.json').read())
print('plain: ms per call', j['ms_per_step'], 'traj/s', j['value'])
rows = list(csv.DictReader(open('gpurun_out/ss/kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print('kernel time total %.1f ms over %d launches' % (tot / 1e6, calls))
for r in rows[:14]:
    print('%8d %10.1f us avg %6.2f %%  %s' % (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['Percentage']), r['Name'][:70]))
PY
