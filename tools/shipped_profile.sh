#!/bin/bash
# predict_ret_best() at the reference's shipped operating point (N=3, K=100 -> 15, H=8, 2 denoise steps): wall time with its
# host / device split, then the per-kernel averages of the same loop under rocprofv3.   JMID_PREC=f16mx tools/shipped_profile.sh
export TMPDIR=/tmp
export JMID_PREC=${JMID_PREC:-f16x3}
python tools/forecaster_latency.py shipped 200
rm -rf gpurun_out/shp; timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/shp -- python tools/forecaster_latency.py shipped 50 > gpurun_out/shp.log 2>&1
f=$(find gpurun_out/shp -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
calls = 51.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per predict_ret_best(): %.1f us" % (tot / calls / 1e3))
for d in rows[:18]:
    print(f"{d['Name'][:86]:86s} per call {float(d['Calls']) / calls:5.1f} x {float(d['AverageNs']) / 1e3:7.2f} us = {float(d['TotalDurationNs']) / calls / 1e3:7.1f} us")
PY
rm -rf gpurun_out/shp
