#!/bin/bash
# SQ issue / wait counters of the big kernels over ONE whole predictor call on one 51-episode chunk (tools/step_only.py): what the
# waves of attention and of the GEMMs spend their cycles on, from the hardware's own counters (the cycle stamps of
# tools/attn_trace.hip say the same from inside the kernel).  One counter per rocprofv3 pass (--kernel-trace + --pmc only).
#   JMID_PREC=f16mx tools/sq_counters.sh      -> gpurun_out/sq/sq_counters_<mode>.json
export TMPDIR=/tmp
export JMID_PREC=${JMID_PREC:-f16mx}
O=gpurun_out/sq; mkdir -p $O
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F8 GRBM_GUI_ACTIVE"
for c in $CNT; do
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- python tools/step_only.py 51 > $O/run_$c.log 2>&1 || { echo "pass $c failed"; tail -3 $O/run_$c.log; continue; }
done
python - $CNT <<'PY'
import collections, csv, glob, json, os, sys
mode = os.environ["JMID_PREC"]
names = sys.argv[1:]
val = {}
for c in names:
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"gpurun_out/sq/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                a = acc[r["Kernel_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    val[c] = acc
kernels = {}
for k, (n, _) in val.get("SQ_WAVE_CYCLES", {}).items():
    if n < 50:
        continue
    row = {"launches": n}
    for c in names:
        if k in val[c]:
            row[c] = val[c][k][1] / val[c][k][0]
    w, wc = row.get("SQ_WAVES"), row.get("SQ_WAVE_CYCLES")
    if w and wc:
        # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); instructions are per wave
        d = {"wave_quad_cycles_per_wave": wc / w}
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU"):
            if c in row:
                d[c.replace("SQ_INSTS_", "insts_") .lower() + "_per_wave"] = row[c] / w
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY"):
            if c in row:
                d[c.lower() + "_frac_of_wave_cycles"] = row[c] / wc
        row["derived"] = d
    kernels[k[:110]] = row
out = {"_comment": "rocprofv3 --kernel-trace --pmc <one counter per pass> -- python tools/step_only.py 51 (one whole predictor call on a "
                   "51-episode chunk, production kernels); per-launch averages of kernels with >= 50 launches; SQ_WAVE_CYCLES, "
                   "SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles",
       "precision": mode, "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0) * kv[1]["launches"]))}
json.dump(out, open(f"gpurun_out/sq/sq_counters_{mode}.json", "w"), indent=1)
for k, v in list(out["kernels"].items())[:6]:
    print(k[:90]); print("   ", {a: round(b, 3) for a, b in v.get("derived", {}).items()})
PY
for c in $CNT; do rm -rf $O/$c; done
