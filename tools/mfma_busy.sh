#!/bin/bash
# MFMA-busy of every kernel of one whole predictor call (51 episodes): SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over
# the 1024 SIMDs) against GRBM_GUI_ACTIVE (shader-clock cycles of the dispatch; rocprofv3 reports the sum over the 8 XCDs:
# the attention launch reads 7.2 M for a ~0.45 ms kernel = 8 x 0.9 M cycles at ~2 GHz), separate --pmc passes.
export TMPDIR=/tmp
export JMID_PREC=${JMID_PREC:-f16x2}
O=gpurun_out/busy; mkdir -p $O
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- python tools/step_only.py 51 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections, json
val = {}
for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"gpurun_out/busy/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                a = acc[r["Kernel_Name"][:70]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    val[c] = acc
rows = []
for k, (n, busy) in val["SQ_VALU_MFMA_BUSY_CYCLES"].items():
    act = val["GRBM_GUI_ACTIVE"].get(k, [0, 0.0])[1]
    if act > 0 and busy > 0:
        rows.append((busy, k, n, busy / (1024.0 * act / 8.0), busy / n, act / n))
rows.sort(reverse=True)
tot_busy = sum(r[0] for r in rows); tot_act = sum(v[1] for v in val["GRBM_GUI_ACTIVE"].values())
out = {"whole_call_mfma_busy": tot_busy / (1024.0 * tot_act / 8.0), "kernels": {r[1]: {"launches": r[2], "mfma_busy": round(r[3], 4), "busy_cycles_per_launch": r[4], "gui_active_per_launch": r[5]} for r in rows[:10]}}
import os
out["precision"] = os.environ["JMID_PREC"]
json.dump(out, open(f"gpurun_out/busy/mfma_busy_{os.environ['JMID_PREC']}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/SQ_VALU_MFMA_BUSY_CYCLES $O/GRBM_GUI_ACTIVE
