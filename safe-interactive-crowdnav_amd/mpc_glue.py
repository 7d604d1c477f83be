"""Caller-side glue between ``predict_ret_best()`` and the MPC parameter layout.

Restates what ``SICNavAcados.predict`` does with the forecaster's result before it reaches the solver
(``sicnav_diffusion/policy/sicnav_acados.py:1644-1666``): drop the prepended current pose, lay the samples out as
``[t, (human sample), xy]`` cut to the MPC horizon (the per-stage parameter blocks of ``select_action``,
``:1389-1395``), pick the initial sample weights, and estimate each human's goal point and preferred speed from the
samples.  Pure NumPy on the host; the MPC itself stays untouched.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

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
    # mean over the samples with the innermost axis reduced, as the reference's per-human np.mean of a 1-D slice
    # does (pairwise summation): the same bits, not just the same value
    goal = np.ascontiguousarray(forecasts[:, :, 0, :].transpose(0, 2, 1)).mean(axis=-1)
    speed = np.linalg.norm(np.diff(forecasts, axis=2), axis=3) / time_step      # [N, k, H-1]
    v_pref = speed.reshape(N, -1).max(axis=1)
    return MPCForecastInputs(forecasts, np.ascontiguousarray(by_stage), init_w, goal, v_pref)


def human_headings(vx: np.ndarray, vy: np.ndarray) -> np.ndarray:
    """theta of the rebuilt FullState list (:1677): atan2(vy, vx), 0 for a human at rest."""
    vx, vy = np.asarray(vx, np.float64), np.asarray(vy, np.float64)
    return np.where((vx != 0) | (vy != 0), np.arctan2(vy, vx), 0.0)


def stage_parameter_blocks(samples_by_stage: np.ndarray, goal_states: np.ndarray, goal_actions: np.ndarray,
                           Q_diag: np.ndarray, R_diag: np.ndarray, term_Q_diag: np.ndarray, horiz: int,
                           static_obs: Optional[np.ndarray] = None) -> np.ndarray:
    """The Acados per-stage parameter vectors of ``select_action`` (``sicnav_acados.py:1388-1413``) as ONE array
    ``[horiz + 1, n_p]``: row i is what the caller passes to ``solver.set(i, "p", ...)``.

        p_i = [goal_states[:, i], goal_actions[:, i], Q_diag, R_diag, term_Q_diag,
               x of the N*k MID samples at stage i, y at stage i, x at stage i+1, y at stage i+1, (static obstacles)]
    for i < horiz; the terminal row takes goal_states[:, horiz], goal_actions[:, horiz-1] and the samples of stages
    horiz-1 / horiz (``:1406-1411``).  ``samples_by_stage`` is ``MPCForecastInputs.samples_by_stage``; like the
    reference this needs horiz + 1 forecast stages (``IndexError`` otherwise).  ``static_obs`` [n_obs, 4] is appended
    flattened in the outdoor setting (``:1382-1383``).
    """
    S = np.asarray(samples_by_stage, np.float64)
    if S.shape[0] < horiz + 1:
        raise IndexError(f"index {S.shape[0]} is out of bounds for axis 0 with size {S.shape[0]}: the MPC horizon "
                         f"{horiz} needs {horiz + 1} forecast stages")
    gs, ga = np.asarray(goal_states, np.float64), np.asarray(goal_actions, np.float64)
    stage = np.arange(horiz + 1)
    s_idx = np.minimum(stage, horiz - 1)                     # samples: terminal row reuses stages horiz-1 / horiz
    a_idx = np.minimum(stage, horiz - 1)
    cost = np.concatenate([Q_diag, R_diag, term_Q_diag]).astype(np.float64)
    cols = [gs[:, stage].T, ga[:, a_idx].T, np.broadcast_to(cost, (horiz + 1, cost.size)),
            S[s_idx, :, 0], S[s_idx, :, 1], S[s_idx + 1, :, 0], S[s_idx + 1, :, 1]]
    if static_obs is not None:
        so = np.asarray(static_obs, np.float64).reshape(-1)
        cols.append(np.broadcast_to(so, (horiz + 1, so.size)))
    return np.ascontiguousarray(np.concatenate(cols, axis=1))
