#!/bin/bash
# per-kernel average durations of a 51-episode f16mx call with and without a tuning knob:  tools/mx_kernel_times.sh "attn_mx=0" "attn_mx=2"
cd /tmp && export TMPDIR=/tmp
for k in "$@"; do
  rm -rf /tmp/kt_$k; 
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$k -- python $GRAFT_REPO_ROOT/tools/mx_mode_probe.py 51 lanes=1 $k > /tmp/kt_$k.log 2>&1
  f=$(find /tmp/kt_$k -name "*kernel_stats.csv" | head -1)
  echo "== $k"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print(f'{r["Name"][:100]:100s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:8.1f} us  total {float(r["TotalDurationNs"])/1e6:8.1f} ms')
PY
done
