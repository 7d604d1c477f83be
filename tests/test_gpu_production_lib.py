"""The PRODUCTION flavour of the library (csrc/libjmid_hip.so: no jmid_dbg_* entry points, no experiment knobs) - what the
drop-in predictor, bench.py and smoke() load - in a process of its own (the test session itself runs on the diagnostics flavour,
tests/conftest.py): the driver's smoke() and a reference golden through the class surface."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROD_ENV = {k: v for k, v in os.environ.items() if k != "JMID_LIB"}

CODE = """
import os, sys, numpy as np, torch
sys.path.insert(0, {repo!r})
import __graft_entry__ as g
from safe_interactive_crowdnav_amd import _lib
from safe_interactive_crowdnav_amd.engine import JmidEngine, JmidError
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
lib = _lib.load_library()
assert not lib.has_diagnostics and b"diagnostics" not in lib.jmid_version() and not hasattr(lib, "jmid_dbg_gemm")
g.smoke()
z = np.load(os.path.join({repo!r}, "tests", "golden", "net_jmid_w256_a5k20t12_s50.npz"))
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), int(z["wseed"])), joint=True, step=50)
for prec in ("f16mx", "f16x2", "f16x3", "f32"):
    vel, _ = eng.denoise(z["x_T"][None], z["ctx"][None], precision=prec, want_pos=False)
    ade = float(np.linalg.norm(vel[0] - z["vel"], axis=-1).mean())
    print(prec, ade)
    assert ade <= 1e-4, (prec, ade)
eng.set_tuning("lanes", 1)
try:
    eng.set_tuning("ln_fuse", 2)
    raise SystemExit("the production library accepted an experiment knob")
except JmidError as e:
    assert e.code == -1
print("PRODUCTION_LIB_OK")
"""


def test_production_library_smoke_and_reference_golden():
    out = subprocess.run([sys.executable, "-c", CODE.format(repo=REPO)], capture_output=True, text=True, timeout=600, cwd=REPO,
                         env=PROD_ENV)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "PRODUCTION_LIB_OK" in out.stdout
