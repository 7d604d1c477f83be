"""Parity margin beyond random-init weights (tools/robustness_sweep.py): weight scale, LayerNorm gain, heavy-tailed weights and
peaked attention on the two shapes the MPC issues, every mode against the oracle in float64 (reference:
DiffusionTraj.sample_sicnav_inference, sicnav_diffusion/JMID/MID/models/diffusion.py:478-541).

Asserted (and quoted in INTEGRATION.md):
  * exact fp32 and f16x3 (the class default) stay within 1e-5 m of the fp64 truth in EVERY cell where f16x3 does not report
    JMID_ERANGE - and when it does, the class falls back to exact fp32 (forecaster.denoise_with_fallback);
  * the opt-in modes stay inside the 1e-4 m gate in the cells INTEGRATION.md lists as their envelope;
  * ``self_check=True`` (the class default since round 6) moves "f16mx" / "f16x2" to f16x3 in every cell where the mode is more than
    ``self_check_tol`` (5e-5 m, half the gate) from f16x3 on the call's own inputs, and leaves it alone where it is within half of that.
"""
import os
import sys
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))

import robustness_sweep as RS                                                 # noqa: E402
from safe_interactive_crowdnav_amd import forecaster as FC                    # noqa: E402

# cells in which the opt-in modes are REQUIRED to hold the gate (their documented envelope); the others are recorded, not required
_INSIDE = ("default", "w_x2", "w_x4", "ln_gain", "ln_gain_w_x2", "student_t", "qk_x4", "qk_x8")     # measured <= 5.3e-5 m (profiles/r04_robustness.json)
ENVELOPE = {"f16x2": _INSIDE, "f16mx": _INSIDE}       # outside (recorded, not required): w_x8 (1.1e-4 ... 3.0e-4 m), qk_x16 (6.5e-5 ... 1.6e-4 m)
_RECORDS = {}


def _cell(name):
    if name not in _RECORDS:
        _RECORDS[name] = RS.run_cell(name)
    return _RECORDS[name]


@pytest.mark.expects_erange
@pytest.mark.parametrize("cell", RS.CELLS)
def test_modes_against_fp64_truth_under_stress(cell):
    rec, _ = _cell(cell)
    for sname, d in rec["shapes"].items():
        m = d["modes"]
        print(cell, sname, "logit spread %.1f nats" % d["layer0_logit_spread_nats"],
              {k: ("ERANGE" if v.get("erange") else "%.2e" % v["ade_vs_fp64_m"]) for k, v in m.items()})
        assert not m["f32"].get("erange") and m["f32"]["ade_vs_fp64_m"] <= 1e-5, (cell, sname, m["f32"])
        if not m["f16x3"].get("erange"):
            assert m["f16x3"]["ade_vs_fp64_m"] <= 1e-5, (cell, sname, m["f16x3"])
        for mode, cells in ENVELOPE.items():
            if cell in cells:
                assert not m[mode].get("erange") and m[mode]["ade_vs_fp64_m"] <= 1e-4, (cell, sname, mode, m[mode])


@pytest.mark.parametrize("cell", RS.EXTREME_CELLS)
def test_no_mode_leaves_the_fp16_range_even_at_absurd_weight_scales(cell):
    """Encoder-layer weights x 16 / x 32 (layer-0 logit spread 300 / 1200 nats - far beyond a trained net; exact fp32 itself is
    1e-5 ... 1e-4 m from the fp64 truth there): NO mode reports JMID_ERANGE (no expects_erange marker: the autouse fixture fails
    the test if the exact-fp32 rerun fires), and f16x3 stays inside the gate, as close to the truth as exact fp32 is.  The range
    cliff the round-4 review asked a per-tensor prescale for does not exist on this net: LayerNorm bounds the residual stream, the
    weights enter the split planes as W * 2^8 with 2^7 of headroom left, and Q arrives pre-scaled."""
    rec, _ = _cell(cell)
    for sname, d in rec["shapes"].items():
        m = d["modes"]
        print(cell, sname, {k: ("ERANGE" if v.get("erange") else "%.2e" % v["ade_vs_fp64_m"]) for k, v in m.items()})
        for mode in RS.MODES:
            assert not m[mode].get("erange"), (cell, sname, mode)
        assert m["f16x3"]["ade_vs_fp64_m"] <= 1e-4, (cell, sname, m["f16x3"])
        assert m["f16x3"]["ade_vs_fp64_m"] <= max(1e-5, 4.0 * m["f32"]["ade_vs_fp64_m"]), (cell, sname, m["f16x3"], m["f32"])


@pytest.mark.expects_erange
@pytest.mark.parametrize("cell", RS.CELLS)
def test_self_check_downgrades_exactly_where_the_mode_drifts(cell, tmp_path):
    rec, w = _cell(cell)
    sname = "shipped"
    d = rec["shapes"][sname]
    sh = RS.SHAPES[sname]
    for mode in ("f16mx", "f16x2"):
        if d["modes"][mode].get("erange") or d["modes"]["f16x3"].get("erange"):
            continue
        delta = d["modes"][mode]["delta_vs_f16x3_m"]
        env, yp = FC.write_configs(str(tmp_path / f"{cell}_{mode}"), joint=True, ctx_dim=256, N=sh["A"], K=sh["K"], k_ret=15, H=sh["T"],
                                   step=sh["step"], time_step=0.25)
        f = FC.HumanTrajectoryForecasterSim(env, yp, weights=w, precision=mode, self_check=True)
        x = d["inputs"]["x_T"][None]
        c = d["inputs"]["ctx"][None]
        p = d["inputs"]["p0"][None]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with f._engine_lock:
                out = f._self_check(x, c, p, d["pos"][mode][None])
        assert abs(f.self_check_delta - delta) <= 1e-7 + 1e-3 * delta       # the same quantity the sweep recorded
        assert f.self_check_tol == 5e-5
        if delta > f.self_check_tol:
            assert f.precision == "f16x3" and not f.self_check, (cell, mode, delta)
            np.testing.assert_array_equal(out[0], d["pos"]["f16x3"])          # the call returns the f16x3 result
        elif delta <= 0.5 * f.self_check_tol:
            assert f.precision == mode, (cell, mode, delta)
