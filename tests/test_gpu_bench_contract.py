"""bench.py prints ONE JSON line with the fields the driver and the judge read (contract in the task statement)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                          "--episodes-per-gpu", "6", "--chunk", "2", "--cpu-episodes", "1"],
                         capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 1 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["data"] == "synthetic" and j["value"] > 0
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = j["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert j["parity"]["pass"] is True and j["parity"]["mean_ADE_vs_oracle_m"] <= 1e-4
    # the default run is the two-term mode; the three-term mode is timed and checked against the oracle in the same run
    assert j["config"]["precision"] == "f16x2" and j["parity"]["precision"] == "f16x2"
    o = j["other_modes"]["f16x3"]
    assert o["value"] > 0 and o["mean_ADE_vs_oracle_m"] <= 1e-5 and o["mean_ADE_between_modes_m"] <= 1e-4
