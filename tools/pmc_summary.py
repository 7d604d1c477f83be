"""Per-kernel PMC summary of ONE whole predictor call (tools/step_only.py) under rocprofv3: what tools/pmc_call.sh writes to
profiles/rNN_pmc_call_<mode>.json, as a function - bench.py calls it after its timed region so that roofline.traffic /
mfma_busy / hbm come from counters measured in the same run on the same box (falling back to the committed file).

    FETCH_SIZE, WRITE_SIZE                       HBM-side bytes (FETCH_SIZE doubled, MI355X_MICROARCH.md HBM section)
    SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE    MFMA-busy fraction per dispatch (busy / (1024 SIMDs x active / 8 XCDs))
each counter in its OWN rocprofv3 pass (--kernel-trace + --pmc only).   python tools/pmc_summary.py f16mx [episodes] -> JSON on stdout
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTERS = ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")


def collect(mode="f16mx", episodes=51, timeout_s=240, keep_dir=None):
    """-> summary dict (same keys as profiles/rNN_pmc_call_<mode>.json, plus "_command"), or None when rocprofv3 is missing or
    a pass fails / times out (a process that aborts under rocprofv3 hangs in its signal handler: every pass is time-limited)."""
    rp = shutil.which("rocprofv3")
    if rp is None:
        return None
    work = keep_dir or tempfile.mkdtemp(prefix="jmid_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", JMID_PREC=mode)
    env.pop("JMID_LIB", None)                   # production kernels
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    val = {}
    cmd_txt = f"rocprofv3 --kernel-trace --pmc <one of {' / '.join(COUNTERS)} per pass> --output-format csv -- python tools/step_only.py {episodes}"
    try:
        for c in COUNTERS:
            d = os.path.join(work, c)
            cmd = [rp, "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.join(REPO, "tools", "step_only.py"), str(episodes)]
            try:
                p = subprocess.run(cmd, cwd=REPO, env=env, timeout=timeout_s, capture_output=True, text=True, start_new_session=True)
            except subprocess.TimeoutExpired:
                return None
            if p.returncode != 0:
                return None
            acc = collections.defaultdict(lambda: [0, 0.0])
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") == c:
                        a = acc[r["Kernel_Name"]]
                        a[0] += 1
                        a[1] += float(r["Counter_Value"])
            if not acc:
                return None
            val[c] = acc
    finally:
        if keep_dir is None:
            shutil.rmtree(work, ignore_errors=True)
    kernels = {}
    for k, (n, fetch) in val["FETCH_SIZE"].items():
        write = val["WRITE_SIZE"].get(k, [0, 0.0])[1]
        busy = val["SQ_VALU_MFMA_BUSY_CYCLES"].get(k, [0, 0.0])[1]
        act = val["GRBM_GUI_ACTIVE"].get(k, [0, 0.0])[1]
        kernels[k[:110]] = {"name": k[:200], "launches": n, "FETCH_SIZE_KB_per_launch": fetch / n, "WRITE_SIZE_KB_per_launch": write / n,
                            "hbm_bytes_per_launch": (2 * fetch + write) * 1024 / n,
                            "mfma_busy": round(busy / (1024.0 * act / 8.0), 4) if act > 0 else None}
    tot_f = sum(v[1] for v in val["FETCH_SIZE"].values())
    tot_w = sum(v[1] for v in val["WRITE_SIZE"].values())
    tot_b = sum(v[1] for v in val["SQ_VALU_MFMA_BUSY_CYCLES"].values())
    tot_a = sum(v[1] for v in val["GRBM_GUI_ACTIVE"].values())
    traj = episodes * 5 * 20
    return {"_comment": "rocprofv3 --kernel-trace --pmc <one counter per pass> -- python tools/step_only.py %d : one whole predictor call "
                        "(encoder -> 50 DDIM steps -> integrator) on one %d-episode chunk = %d trajectories, production kernels; "
                        "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB counters; FETCH_SIZE doubled per MI355X_MICROARCH.md); "
                        "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)" % (episodes, episodes, traj),
            "_command": cmd_txt,
            "precision": mode, "chunk_episodes": episodes, "tokens": episodes * 5 * 20 * 12, "trajectories": traj,
            "whole_call_mfma_busy": round(tot_b / (1024.0 * tot_a / 8.0), 4) if tot_a else None,
            "call": {"FETCH_SIZE_KB_total": tot_f, "WRITE_SIZE_KB_total": tot_w, "hbm_bytes_per_call": (2 * tot_f + tot_w) * 1024,
                     "hbm_bytes_per_trajectory": int((2 * tot_f + tot_w) * 1024 / traj)},
            "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))}


if __name__ == "__main__":
    out = collect(sys.argv[1] if len(sys.argv) > 1 else "f16mx", int(sys.argv[2]) if len(sys.argv) > 2 else 51)
    if out is None:
        raise SystemExit("PMC collection failed (rocprofv3 missing, or a pass failed / timed out)")
    out.pop("_command", None)
    print(json.dumps(out, indent=1))
