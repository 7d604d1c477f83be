"""Caller-side glue between ``predict_ret_best()`` and the MPC parameter layout.

Restates what ``SICNavAcados.predict`` does with the forecaster's result before it reaches the solver
(``sicnav_diffusion/policy/sicnav_acados.py:1644-1666``): drop the prepended current pose, lay the samples out as
``[t, (human sample), xy]`` cut to the MPC horizon (the per-stage parameter blocks of ``select_action``,
``:1389-1395``), pick the initial sample weights, and estimate each human's goal point and preferred speed from the
samples.  Pure NumPy on the host; the MPC itself stays untouched.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class MPCForecastInputs:
    forecasts: np.ndarray          # [N, k, H, 2]   forecasts without the t0 pose            (:1645)
    samples_by_stage: np.ndarray   # [min(H, horiz+1), N*k, 2]  'h s t d -> t (h s) d'        (:1651)
    init_weights: np.ndarray       # [k] (joint) or [N, k] (independent)                      (:1646-1649)
    goal_xy: np.ndarray            # [N, 2] mean first-step position over samples             (:1661-1662)
    v_pref: np.ndarray             # [N]    max finite-difference speed over samples, steps   (:1667)


def mpc_forecast_inputs(top_k_forecasts: np.ndarray, top_k_weights: np.ndarray, horiz: int, time_step: float,
                        joint: bool) -> MPCForecastInputs:
    forecasts = top_k_forecasts[:, :, 1:, :]
    init_w = top_k_weights[0, :] if joint else top_k_weights
    N, k, H, _ = forecasts.shape
    by_stage = forecasts.transpose(2, 0, 1, 3).reshape(H, N * k, 2)[: horiz + 1]
    goal = forecasts[:, :, 0, :].mean(axis=1)
    speed = np.linalg.norm(np.diff(forecasts, axis=2), axis=3) / time_step      # [N, k, H-1]
    v_pref = speed.reshape(N, -1).max(axis=1)
    return MPCForecastInputs(forecasts, np.ascontiguousarray(by_stage), init_w, goal, v_pref)
