#!/bin/bash
# Per-kernel PMC summary of ONE single-scene predictor call (tools/step_only.py 1): where the bytes of a small-M launch come
# from.  FETCH_SIZE / WRITE_SIZE (fabric side of the L2s) and TCC_HIT / TCC_MISS, each in its own rocprofv3 pass, plus the
# kernel-trace averages of the same call.   JMID_PREC=f16mx tools/small_pmc.sh [episodes]   -> gpurun_out/small_pmc_<mode>_E<e>.txt
export TMPDIR=/tmp
export JMID_PREC=${JMID_PREC:-f16mx}
E=${1:-1}
O=gpurun_out/spmc; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=$O/$(echo $c | tr ' ' '_')
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python tools/step_only.py $E > $d.log 2>&1 || { echo "pass $c failed"; tail -3 $d.log; exit 1; }
done
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python tools/step_only.py $E > $O/stats.log 2>&1
python - $E <<'PY' | tee gpurun_out/small_pmc_${JMID_PREC}_E$E.txt
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/spmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
dur = {}
for f in glob.glob("gpurun_out/spmc/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
print("episodes", sys.argv[1])
print("%-70s %6s %8s %9s %9s %7s" % ("kernel", "calls", "avg us", "fetch KB", "write KB", "L2 hit"))
for k, (n, us) in sorted(dur.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    c = acc.get(k, {})
    f = c.get("FETCH_SIZE", [1, 0])
    w = c.get("WRITE_SIZE", [1, 0])
    h = c.get("TCC_HIT_sum", [1, 0])[1]
    m = c.get("TCC_MISS_sum", [1, 0])[1]
    print("%-70s %6d %8.2f %9.1f %9.1f %7.3f" % (k[:70], n, us, 2 * f[1] / max(f[0], 1), w[1] / max(w[0], 1), h / max(h + m, 1)))
PY
rm -rf $O
