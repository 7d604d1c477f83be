#!/bin/bash
# rocprofv3 per-kernel averages of a 51-episode call in one mode under knob settings:  tools/kernel_times.sh f16x2 "vt_stage=0" "vt_stage=3"
cd /tmp && export TMPDIR=/tmp
prec=$1; shift
for k in "$@"; do
  rm -rf /tmp/kt2_$k
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2_$k -- python $GRAFT_REPO_ROOT/tools/ab_knob.py $prec lanes 1 1 51 $k > /tmp/kt2_$k.log 2>&1
  f=$(find /tmp/kt2_$k -name "*kernel_stats.csv" | head -1)
  echo "== $prec $k"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:8]:
    print(f'{r["Name"][:100]:100s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
done
