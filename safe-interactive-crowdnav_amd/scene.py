"""Host-side batch builder: agent histories -> the arrays the device encoder consumes.

A NumPy restatement (no pandas, no Scene/Node objects) of what the reference does between
``update_state_hists`` and ``Trajectron.get_latent``:

  * history frame table, resampling to the ``time_step`` grid from the last stamp backwards, keep-last per
    bin, linear interpolation of empty bins      (``JMID/mid_sim_wrapper.py:244-310``)
  * cluster selection around the human nearest the robot, constant-velocity forecasts for everybody else
                                                 (``mid_sim_wrapper.py:313-437``)
  * per-node state [pos, vel, acc] by first differences   (``MID/environment/data_utils.py:24-37``)
  * standardisation, neighbour sets, edge scaling (``MID/dataset/preprocessing.py:428-620``,
    ``MID/environment/scene_graph.py:111-250, 280-313``)
  * the reductions the encoder applies before its LSTMs: per edge type the SUM of neighbour histories and
    clamp(sum(edge values), 1)                    (``MID/models/encoders/mgcvae.py:726-768``)

All arithmetic is float64 until the final cast to float32, as in the reference
(``preprocessing.py:495-499``: ``torch.tensor(..., dtype=torch.float)``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

ATTENTION_RADIUS = 3.0                       # mid_sim_wrapper.py:234-238
STATE_STD = np.array([3.0, 3.0, 2.0, 2.0, 1.0, 1.0])   # pos std overridden by the attention radius
                                                        # (mid_sim_wrapper.py:219-231, preprocessing.py:477-478)
EDGE_ADDITION_FILTER = (0.25, 0.5, 0.75, 1.0)           # trajectron_hypers.py:83
EDGE_REMOVAL_FILTER = (1.0, 0.0)                        # trajectron_hypers.py:84
TYPE_VALUE_PED, TYPE_VALUE_ROBOT = 1, 2                 # NodeType enum values (environment/node_type.py)
ROBOT_ID = -1


class HistoryTooShortError(TypeError):
    """Fewer resampled frames than ``past_num_frames``.  The reference fails in the same situation
    (``get_timesteps_data`` returns None -> ``TypeError`` at MID/mid.py:326); callers catch and re-raise
    (``sicnav_diffusion/policy/sicnav_acados.py:1176-1180``)."""


def derivative_of(x: np.ndarray, dt: float) -> np.ndarray:
    """data_utils.py:24-37 for NaN-free input: first difference, first element duplicated."""
    if x.shape[0] < 2:
        return np.zeros_like(x)
    d = np.empty_like(x)                 # (x: [F] or [F, C], differences along the first axis; the values np.ediff1d(x, to_begin=x[1] - x[0]) gives)
    np.subtract(x[1:], x[:-1], out=d[1:])
    d[0] = d[1]
    return d / dt


# --------------------------------------------------------------------------------------------- history table
def frame_table(prev_states: Sequence[Sequence[Sequence[float]]], prev_robot_states: Sequence[Sequence[float]],
                time_step: float, num_hist_frames: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """mid_sim_wrapper.py:244-298.

    prev_states: per human a list of [x, y, t]; prev_robot_states: list of [x, y, t].
    Returns (human_xy [F, N, 2], robot_xy [F, 2], pose_now [N, 2]) with F <= num_hist_frames resampled frames.
    """
    N = len(prev_states)
    h0 = np.asarray(prev_states[0], dtype=np.float64).reshape(-1, 3)
    times = h0[:, 2]
    table = np.full((len(times), 2 * N + 2), np.nan)
    table[:, 0:2] = h0[:, 0:2]
    for i in range(1, N):                       # left-join on exactly equal stamps
        hi = np.asarray(prev_states[i], dtype=np.float64).reshape(-1, 3)
        lut = {t: k for k, t in enumerate(hi[:, 2])}
        for r, t in enumerate(times):
            k = lut.get(t)
            if k is not None:
                table[r, 2 * i:2 * i + 2] = hi[k, 0:2]
    pose_now = table[-1, :2 * N].reshape(N, 2).copy()      # agent_df.tail(1), before the robot join
    rob = np.asarray(prev_robot_states, dtype=np.float64).reshape(-1, 3)
    lut = {t: k for k, t in enumerate(rob[:, 2])}
    for r, t in enumerate(times):
        k = lut.get(t)
        if k is not None:
            table[r, 2 * N:2 * N + 2] = rob[k, 0:2]
    keep = ~np.isnan(table).any(axis=1)         # dropna
    table, times = table[keep], times[keep]
    order = np.argsort(times, kind="stable")
    table, times = table[order], times[order]
    if len(times) == 0:
        raise HistoryTooShortError("no complete history frame")
    # subsample_df: stamps*100 -> integer nanoseconds (truncation), bins of round(time_step*100) ns anchored at
    # the LAST stamp, right-closed; keep the last row per bin; interpolate empty bins linearly
    ns = np.trunc(times * 100.0).astype(np.int64)
    w = int(round(time_step * 100))
    k = (ns[-1] - ns) // w                      # bin index counted backwards from the end
    nb = int(k.max()) + 1
    out = np.full((nb, table.shape[1]), np.nan)
    for r in range(len(times)):                 # ascending time: later rows overwrite -> "last"
        out[nb - 1 - int(k[r])] = table[r]
    idx = np.arange(nb)
    good = ~np.isnan(out[:, 0])
    if not good.all():
        for c in range(out.shape[1]):
            out[:, c] = np.interp(idx, idx[good], out[good, c])
    out = out[-num_hist_frames:]
    human_xy = out[:, :2 * N].reshape(-1, N, 2)
    robot_xy = out[:, 2 * N:2 * N + 2]
    return human_xy, robot_xy, pose_now


# --------------------------------------------------------------------------------------------- scene batch
@dataclass
class SceneBatch:
    ids_in: np.ndarray               # pedestrian track ids inside the chosen cluster, ascending == batch row order
    ids_out: np.ndarray              # pedestrian track ids outside the cluster (constant-velocity fill)
    x: np.ndarray                    # [A, F, 6] f32 raw state  [px, py, vx, vy, ax, ay]
    x_st: np.ndarray                 # [A, F, 6] f32 standardized
    nbr_sum: np.ndarray              # [A, 2, F, 6] f32   edge types (PED->PED, PED->ROBOT)
    edge_mask: np.ndarray            # [A, 2] f32
    p0: np.ndarray                   # [A, 2] f32 current positions (integrator initial condition)
    cv_forecasts: Dict[int, np.ndarray] = field(default_factory=dict)   # id -> [H, 2] f64
    pose_now: Optional[np.ndarray] = None                               # [N, 2] f64
    robot_in_cluster: bool = False


def edge_scaling_last(adj3: np.ndarray) -> np.ndarray:
    """scene_graph.py:203-225 evaluated at the last of 3 frames.  adj3 [3, n, n] (type-valued adjacency).
    conv with the addition filter, clamp 1, zero where not adjacent now; the removal filter [1, 0] is the
    identity at this frame."""
    f = EDGE_ADDITION_FILTER
    s = np.minimum(f[0] * adj3[2] + f[1] * adj3[1] + f[2] * adj3[0], 1.0)
    s = np.where(adj3[2] == 0, 0.0, s)
    return np.minimum(s, 1.0)


def build_scene(human_xy: np.ndarray, robot_xy: np.ndarray, time_step: float, horizon: int,
                num_hist_frames: int = 6, force_all_in_cluster: bool = False) -> SceneBatch:
    """mid_sim_wrapper.py:313-437 + preprocessing.py:428-694 for one scene.

    human_xy [F, N, 2], robot_xy [F, 2] on the ``time_step`` grid (oldest first).
    """
    F, N, _ = human_xy.shape
    if F < num_hist_frames:
        raise HistoryTooShortError(f"{F} history frames available, {num_hist_frames} needed")
    dt = time_step
    # positions at the last frame, sorted by track id: robot (-1) first
    pos_last = np.concatenate([robot_xy[-1:], human_xy[-1]], axis=0)           # [N+1, 2]
    track_ids = np.concatenate([[ROBOT_ID], np.arange(N)])
    sq = np.square(pos_last[:, None] - pos_last[None, :])
    dists = np.sqrt(np.sum(sq, axis=2))
    mask = dists < ATTENTION_RADIUS
    if force_all_in_cluster:
        in_mask = np.ones(N + 1, dtype=bool)
    else:
        cluster_means = (mask @ pos_last) / mask.sum(axis=1, keepdims=True)
        robot_dist = np.linalg.norm(cluster_means - pos_last[0], axis=1)
        chosen = int(np.argmin(robot_dist[1:])) + 1
        in_mask = mask[chosen]
    ids_in_all = track_ids[in_mask]
    ids_out_all = track_ids[~in_mask]

    # scene nodes in track-id order (robot first when it is in the cluster); node state [pos, vel, acc] by first differences for all
    # of them at once (element-wise, so the values derivative_of gives per coordinate)
    node_ids = list(ids_in_all)
    n = len(node_ids)
    pos_all = np.concatenate([robot_xy[:, None, :], human_xy], axis=1)         # [F, N+1, 2], robot first like track_ids

    def deriv(a):                                                              # [n, F, 2], differences along the frames
        if F < 2:
            return np.zeros_like(a)
        d = np.empty_like(a)
        np.subtract(a[:, 1:], a[:, :-1], out=d[:, 1:])
        d[:, 0] = d[:, 1]
        return d / dt

    P = np.ascontiguousarray(pos_all[:, in_mask].transpose(1, 0, 2))           # [n, F, 2]
    V = deriv(P)
    S = np.concatenate([P, V, deriv(V)], axis=2)                               # [n, F, 6]
    types = np.where(ids_in_all == ROBOT_ID, TYPE_VALUE_ROBOT, TYPE_VALUE_PED)

    # constant-velocity forecasts for pedestrians outside the cluster (mid_sim_wrapper.py:413-429)
    cv = {}
    out_peds = [int(i) for i in ids_out_all if i != ROBOT_ID]
    if out_peds:
        last = human_xy[-1, out_peds]                                          # [m, 2]
        v_last = (last - human_xy[-2, out_peds]) / dt if F >= 2 else np.zeros_like(last)
        fc_all = last[:, None, :] + np.cumsum(np.repeat((v_last * dt)[:, None, :], horizon, axis=1), axis=1)
        for r, i in enumerate(out_peds):
            cv[i] = fc_all[r]

    # temporal scene graph over the last 3 frames (scene.py:67-110, scene_graph.py:111-201)
    P3 = S[:, F - 3:F, 0:2].transpose(1, 0, 2)                                 # [3, n, 2]
    d3 = np.sqrt(np.square(P3[:, :, None] - P3[:, None, :]).sum(-1))           # [3, n, n]
    type_mat = np.tile(types[None, :], (n, 1)).astype(np.float64)
    np.fill_diagonal(type_mat, 0)
    adj3 = (d3 <= ATTENTION_RADIUS).astype(np.float64) * type_mat[None]
    eye = np.eye(n, dtype=bool)
    adj3[:, eye] = 0.0
    scaling = edge_scaling_last(adj3)                                          # [n, n]
    connected = scaling > 1e-2

    ped_rows = np.nonzero(ids_in_all != ROBOT_ID)[0]
    A = len(ped_rows)
    x = S[ped_rows]                                                            # [A, F, 6]
    rel = np.zeros((A, 1, 6))
    rel[:, 0, 0:2] = x[:, -1, 0:2]
    x_st = (x - rel) / STATE_STD
    nbr_sum = np.zeros((A, 2, F, 6), dtype=np.float32)
    edge_mask = np.zeros((A, 2), dtype=np.float32)
    for r, k in enumerate(ped_rows):
        # edge values are NOT filtered by edge type (scene_graph.py:293-299): both types see the same sum
        ev = scaling[k, connected[k]].astype(np.float32)
        edge_mask[r, :] = np.float32(min(float(ev.sum(dtype=np.float32)), 1.0)) if ev.size else np.float32(0.0)
    # neighbour state relative to the ego's WHOLE present state (preprocessing.py:531-550), summed per edge type in node order in
    # float32: one pass over the neighbours j for all egos at once (a neighbour that is not connected adds +0.0, which leaves a
    # float32 partial sum as it is - the sums start at +0.0 and never become -0.0)
    if A and n:
        conn = connected[ped_rows]                                             # [A, n]
        tm = type_mat[ped_rows]
        sel = np.stack([conn & (tm == TYPE_VALUE_PED), conn & (tm == TYPE_VALUE_ROBOT)], axis=1)   # [A, 2, n]
        ego_now = x[:, -1, :]                                                  # [A, 6]
        for j in range(n):
            if not sel[:, :, j].any():
                continue
            term = ((S[j][None] - ego_now[:, None, :]) / STATE_STD).astype(np.float32)             # [A, F, 6]
            nbr_sum += np.where(sel[:, :, j, None, None], term[:, None], np.float32(0.0))
    ids_in = np.array([i for i in node_ids if i != ROBOT_ID], dtype=np.int64)
    ids_out = np.array(out_peds, dtype=np.int64)
    return SceneBatch(ids_in=ids_in, ids_out=ids_out, x=x.astype(np.float32), x_st=x_st.astype(np.float32),
                      nbr_sum=nbr_sum, edge_mask=edge_mask, p0=x[:, -1, 0:2].astype(np.float32),
                      cv_forecasts=cv, robot_in_cluster=bool(in_mask[0]))


# --------------------------------------------------------------------------------------------- batched builder
def build_scenes_batched(human_xy: np.ndarray, robot_xy: np.ndarray, time_step: float,
                         force_all_in_cluster: bool = False, horizon: Optional[int] = None) -> Dict[str, np.ndarray]:
    """``build_scene`` for E independent episodes at once (no Python loop over episodes): the feed of the
    256-4096-episode evaluation sweeps.

    human_xy [E, F, N, 2], robot_xy [E, F, 2].  Every pedestrian gets a row; ``in_cluster`` [E, N] tells which rows
    the reference would have sent through the network (the others get constant-velocity forecasts there).
    Returns x, x_st [E, N, F, 6], nbr_sum [E, N, 2, F, 6], edge_mask [E, N, 2], p0 [E, N, 2] (all float32),
    in_cluster [E, N] bool, robot_in_cluster [E] bool.  Rows outside the cluster are computed as if they had no
    neighbours in the graph (they are not graph nodes in the reference).  Bit-identical to ``build_scene`` per episode.
    With ``horizon`` the constant-velocity forecasts of EVERY pedestrian (what the reference returns for the ones
    outside the cluster, mid_sim_wrapper.py:413-429) come back as ``cv`` [E, N, horizon, 2] float64.
    """
    E, F, N, _ = human_xy.shape
    dt = time_step
    pos = np.concatenate([robot_xy[:, :, None, :], human_xy], axis=2)              # [E, F, N+1, 2], robot first
    last = pos[:, -1]                                                              # [E, N+1, 2]
    d_last = np.sqrt(np.square(last[:, :, None] - last[:, None, :]).sum(-1))
    near = d_last < ATTENTION_RADIUS
    if force_all_in_cluster:
        inc = np.ones((E, N + 1), dtype=bool)
    else:
        means = (near.astype(np.float64) @ last) / near.sum(axis=2, keepdims=True)
        rdist = np.linalg.norm(means - last[:, :1], axis=2)
        chosen = np.argmin(rdist[:, 1:], axis=1) + 1
        inc = near[np.arange(E), chosen]                                           # [E, N+1]
    # node states [E, N+1, F, 6] by first differences (first element duplicated)
    P = pos.transpose(0, 2, 1, 3)                                                  # [E, N+1, F, 2]

    def deriv(a):
        dd = np.diff(a, axis=2) / dt
        return np.concatenate([dd[:, :, :1], dd], axis=2)

    V = deriv(P)
    A_ = deriv(V)
    S = np.concatenate([P, V, A_], axis=3)                                         # [E, N+1, F, 6]
    # temporal scene graph over the last three frames, in-cluster nodes only
    P3 = P[:, :, F - 3:F].transpose(0, 2, 1, 3)                                    # [E, 3, N+1, 2]
    d3 = np.sqrt(np.square(P3[:, :, :, None] - P3[:, :, None, :]).sum(-1))         # [E, 3, n, n]
    types = np.full(N + 1, float(TYPE_VALUE_PED))
    types[0] = float(TYPE_VALUE_ROBOT)
    tmat = np.tile(types[None, :], (N + 1, 1))
    np.fill_diagonal(tmat, 0)
    pair_in = (inc[:, :, None] & inc[:, None, :]).astype(np.float64)               # [E, n, n]
    adj3 = (d3 <= ATTENTION_RADIUS).astype(np.float64) * tmat[None, None] * pair_in[:, None]
    f = EDGE_ADDITION_FILTER
    scal = np.minimum(f[0] * adj3[:, 2] + f[1] * adj3[:, 1] + f[2] * adj3[:, 0], 1.0)
    scal = np.where(adj3[:, 2] == 0, 0.0, scal)                                    # [E, n, n]
    conn = scal > 1e-2
    xs = S[:, 1:]                                                                   # pedestrians [E, N, F, 6]
    rel = np.zeros((E, N, 1, 6))
    rel[:, :, 0, 0:2] = xs[:, :, -1, 0:2]
    x_st = (xs - rel) / STATE_STD
    em = np.minimum((scal[:, 1:] * conn[:, 1:]).astype(np.float32).sum(axis=2, dtype=np.float32), np.float32(1.0))
    edge_mask = np.repeat(em[:, :, None], 2, axis=2).astype(np.float32)
    nbr_sum = np.zeros((E, N, 2, F, 6), dtype=np.float32)
    ego_now = xs[:, :, -1:, :]                                                      # [E, N, 1, 6]
    for j in range(N + 1):                                                          # neighbours in node order
        relj = ((S[:, j][:, None] - ego_now) / STATE_STD).astype(np.float32)        # [E, N, F, 6]
        w = conn[:, 1:, j].astype(np.float32)[:, :, None, None]
        e_idx = 1 if j == 0 else 0                                                  # robot -> edge type PED->ROBOT
        nbr_sum[:, :, e_idx] = nbr_sum[:, :, e_idx] + relj * w
    out = dict(x=xs.astype(np.float32), x_st=x_st.astype(np.float32), nbr_sum=nbr_sum, edge_mask=edge_mask,
               p0=xs[:, :, -1, 0:2].astype(np.float32), in_cluster=inc[:, 1:], robot_in_cluster=inc[:, 0])
    if horizon is not None:      # same operations, in the same order, as build_scene's per-pedestrian loop
        step = np.repeat((xs[:, :, -1, 2:4] * dt)[:, :, None, :], horizon, axis=2)     # [E, N, H, 2]
        out["cv"] = xs[:, :, -1:, 0:2] + np.cumsum(step, axis=2)
    return out


# --------------------------------------------------------------------------------------------- synthetic feeds
def synthetic_episodes(E: int, N: int, seed: int, time_step: float = 0.25, num_hist_frames: int = 6,
                       horizon: int = 12) -> Dict[str, np.ndarray]:
    """Synthetic scene batches for measurement: per episode pos0 ~ U(-2,2)^2,
    vel ~ U(-0.5,0.5)^2 m/s, 7 frames at dt (6 kept), robot at (0,-3) + 0.2 t y; all agents forced into the
    cluster (A = N).  Returns stacked encoder inputs and the constant-velocity ground truth."""
    rng = np.random.default_rng(seed)
    pos0 = rng.uniform(-2.0, 2.0, (E, N, 2))
    vel = rng.uniform(-0.5, 0.5, (E, N, 2))
    t = np.arange(num_hist_frames + 1) * time_step
    hum = pos0[:, None] + vel[:, None] * t[None, :, None, None]                    # [E, F+1, N, 2]
    rob = np.array([0.0, -3.0])[None, None] + np.array([0.0, 0.2])[None, None] * t[None, :, None]
    rob = np.broadcast_to(rob, (E, num_hist_frames + 1, 2))
    b = build_scenes_batched(hum[:, -num_hist_frames:], rob[:, -num_hist_frames:], time_step, force_all_in_cluster=True)
    steps = (np.arange(horizon) + 1)[None, None, :, None] * time_step
    gt = (hum[:, -1][:, :, None, :] + vel[:, :, None, :] * steps).astype(np.float32)
    return dict(x=b["x"], x_st=b["x_st"], nbr_sum=b["nbr_sum"], edge_mask=b["edge_mask"], p0=b["p0"], gt=gt)
