export TMPDIR=/tmp
for e in 4 8; do
rm -rf gpurun_out/es$e
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/es$e -- python bench.py --precision f16mx --modes f16mx --steps 3 --warmup 1 --cpu-episodes 0 --no-e2e --no-profile --episodes-per-gpu $e > gpurun_out/es$e.log 2>&1
f=$(find gpurun_out/es$e -name "*kernel_stats.csv" | head -1)
echo "== E=$e"; grep '^{' gpurun_out/es$e.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for d in rows[:16]:
    print(f"{d['Name'][:75]:75s} calls {d['Calls']:>5s} avg {float(d['AverageNs'])/1e3:7.1f} us tot {float(d['TotalDurationNs'])/1e6:7.2f} ms {d['Percentage']}")
PY
rm -rf gpurun_out/es$e
done
