#!/bin/bash
# Per-kernel PMC summary of ONE whole predictor call on one 51-episode chunk (tools/step_only.py), production kernels
# only - what bench.py quotes under roofline.traffic / roofline.mfma_busy / hbm (with the file's git blob hash):
#   FETCH_SIZE, WRITE_SIZE                       HBM-side bytes (FETCH_SIZE doubled, MI355X_MICROARCH.md HBM section)
#   SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE    MFMA-busy fraction per dispatch (busy / (1024 SIMDs x active / 8 XCDs))
# each counter in its OWN rocprofv3 pass (--kernel-trace + --pmc only).  Run on the GPU box from the repo root:
#   JMID_PREC=f16x2 tools/pmc_call.sh ; JMID_PREC=f16x3 tools/pmc_call.sh     -> gpurun_out/pmc/pmc_call_<mode>.json
export TMPDIR=/tmp
export JMID_PREC=${JMID_PREC:-f16x2}
O=gpurun_out/pmc; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  # (-k 5: a process that aborts under rocprofv3 hangs in its signal handler; one bad pass must not eat the call's time limit)
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- python tools/step_only.py 51 > $O/run_$c.log 2>&1 || { echo "pass $c failed"; tail -3 $O/run_$c.log; exit 1; }
done
python - <<'PY'
import collections, csv, glob, json, os
mode = os.environ["JMID_PREC"]
val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"gpurun_out/pmc/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                a = acc[r["Kernel_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    val[c] = acc
kernels = {}
for k, (n, fetch) in val["FETCH_SIZE"].items():
    write = val["WRITE_SIZE"].get(k, [0, 0.0])[1]
    busy = val["SQ_VALU_MFMA_BUSY_CYCLES"].get(k, [0, 0.0])[1]
    act = val["GRBM_GUI_ACTIVE"].get(k, [0, 0.0])[1]
    kernels[k[:110]] = {"name": k[:200], "launches": n, "FETCH_SIZE_KB_per_launch": fetch / n, "WRITE_SIZE_KB_per_launch": write / n,
                        "hbm_bytes_per_launch": (2 * fetch + write) * 1024 / n,
                        "mfma_busy": round(busy / (1024.0 * act / 8.0), 4) if act > 0 else None}
tot_f = sum(v[1] for v in val["FETCH_SIZE"].values()); tot_w = sum(v[1] for v in val["WRITE_SIZE"].values())
tot_b = sum(v[1] for v in val["SQ_VALU_MFMA_BUSY_CYCLES"].values()); tot_a = sum(v[1] for v in val["GRBM_GUI_ACTIVE"].values())
out = {"_comment": "rocprofv3 --kernel-trace --pmc <one counter per pass> -- python tools/step_only.py 51 : one whole predictor call "
                   "(encoder -> 50 DDIM steps -> integrator) on one 51-episode chunk = 5100 trajectories, production kernels; "
                   "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB counters; FETCH_SIZE doubled per MI355X_MICROARCH.md); "
                   "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)",
       "precision": mode, "chunk_episodes": 51, "tokens": 61200, "trajectories": 5100,
       "whole_call_mfma_busy": round(tot_b / (1024.0 * tot_a / 8.0), 4) if tot_a else None,
       "call": {"FETCH_SIZE_KB_total": tot_f, "WRITE_SIZE_KB_total": tot_w, "hbm_bytes_per_call": (2 * tot_f + tot_w) * 1024,
                "hbm_bytes_per_trajectory": int((2 * tot_f + tot_w) * 1024 / 5100)},
       "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))}
json.dump(out, open(f"gpurun_out/pmc/pmc_call_{mode}.json", "w"), indent=1)
print(mode, "HBM bytes per trajectory", out["call"]["hbm_bytes_per_trajectory"], "whole-call MFMA busy", out["whole_call_mfma_busy"])
for k, v in list(out["kernels"].items())[:8]:
    print("%6d  %8.1f MB/launch  busy %s  %s" % (v["launches"], v["hbm_bytes_per_launch"] / 1e6, v["mfma_busy"], k[:80]))
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE $O/SQ_VALU_MFMA_BUSY_CYCLES $O/GRBM_GUI_ACTIVE
