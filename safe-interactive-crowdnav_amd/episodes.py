"""Batched episode generator for the multi-episode evaluation sweeps.

The reference produces the histories the predictor sees by stepping ``CrowdSimPlus`` one episode at a time
(``crowd_sim_plus/envs/crowd_sim_plus.py:609-765`` reset, ``:1025-1258`` step), every human choosing its velocity
with ORCA through Python-RVO2 (``crowd_sim_plus/envs/policy/orca.py:82-133``: a fresh simulator per step and per
human, the human as agent 0, everybody else's preferred velocity (0, 0)).  For BASELINE configs 3 and 5 (256 - 4096
parallel episodes) this module does the same for E episodes at once, in NumPy on the host: circle-crossing and
square-crossing placement (``crowd_sim_plus.py:454-481, 484-520``), agent-agent ORCA for every (episode, agent) pair in one vectorised pass
(half-plane construction and the incremental 2-D linear programs of RVO2's ``Agent::computeNewVelocity`` /
``linearProgram1-3``; van den Berg et al., ISRR 2011), holonomic position update.  It stays on the host like the
rest of the simulator: a few thousand 2-D LPs with <= 10 constraints per step are microseconds of work next to the
denoise loop, and the MPC side that consumes the same states lives there too.

PARITY: partially pinned.  Everything the reference itself computes around rvo2 is pinned bit for bit by fixtures that
``tests/golden/make_golden_episodes.py`` generates by executing the reference's own lines against stand-ins
(``tests/golden/episodes_*.npz``): the circle-crossing and the square-crossing placement incl. their draw order from the
generator (``crowd_sim_plus.py:454-481, 484-520``; ``place_circle_crossing_humans``, ``place_square_crossing_humans``), and what is handed to rvo2 per step - simulator and agent
parameters, inflated radii, speed limits, preferred velocities (``orca.py:56-67, 93-129``; ``orca_call_parameters``).
rvo2 ITSELF (RVO2 Library 2.0.2 behind Python-RVO2) STAYS UNPINNED: it is an un-vendored C++ dependency, absent from the
reference tree and from this image, so no output of its ORCA solver exists to compare with.  For that part what is
checked (``tests/test_episodes.py``): equality with a scalar, one-agent-at-a-time restatement of the published
algorithm kept with the test infrastructure, optimality against a brute-force search of the velocity disc,
collision-freeness and goal progress of the generated crowds.  The shipped scenarios of the reference (``hallway*``, ``env.config:16-17``)
- walls, RVO2's obstacle ORCA lines, door sub-goals, wall-constrained actions, and what a ``step()`` decides (collision, goal,
timeout, rewards) - live in ``crowd_env.py`` (``simulate_hallway``), which reuses the linear programs of this module.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np

RVO_EPSILON = 0.00001


def _det(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    return a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]


def _dot(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    return a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]


# ------------------------------------------------------------------------------------------------ ORCA half-planes
def orca_lines(pos, vel, radius, opos, ovel, orad, time_horizon: float, time_step: float, with_dist: bool = False):
    """pos, vel [B, 2], radius [B]; the other agents opos, ovel [B, L, 2], orad [B, L].  Returns (point, direction)
    [B, L, 2] each, neighbours ordered nearest first (RVO2 keeps its neighbour list sorted by distance); with
    ``with_dist`` also their squared distances [B, L] in that order."""
    rp = opos - pos[:, None, :]
    dist_sq = _dot(rp, rp)
    order = np.argsort(dist_sq, axis=1, kind="stable")
    take = lambda a: np.take_along_axis(a, order if a.ndim == 2 else order[..., None], axis=1)
    rp, dist_sq, ovel, orad = take(rp), take(dist_sq), take(ovel), take(orad)
    rv = vel[:, None, :] - ovel
    cr = radius[:, None] + orad
    cr_sq = cr * cr
    apart = dist_sq > cr_sq
    inv_t = np.where(apart, 1.0 / time_horizon, 1.0 / time_step)[..., None]
    w = rv - inv_t * rp
    w_len_sq = _dot(w, w)
    with np.errstate(invalid="ignore", divide="ignore"):
        w_len = np.sqrt(w_len_sq)
        uw = w / w_len[..., None]
        dp1 = _dot(w, rp)
        # projection on the cut-off circle (also the whole collision case, with 1 / time_step)
        circle = ~apart | ((dp1 < 0.0) & (dp1 * dp1 > cr_sq * w_len_sq))
        dir_c = np.stack([uw[..., 1], -uw[..., 0]], axis=-1)
        u_c = (cr * inv_t[..., 0] - w_len)[..., None] * uw
        # projection on a leg of the velocity obstacle
        leg = np.sqrt(np.where(apart, dist_sq - cr_sq, 1.0))
        left = _det(rp, w) > 0.0
        dl = np.stack([rp[..., 0] * leg - rp[..., 1] * cr, rp[..., 0] * cr + rp[..., 1] * leg], axis=-1) / dist_sq[..., None]
        dr = -np.stack([rp[..., 0] * leg + rp[..., 1] * cr, -rp[..., 0] * cr + rp[..., 1] * leg], axis=-1) / dist_sq[..., None]
        dir_l = np.where(left[..., None], dl, dr)
        u_l = _dot(rv, dir_l)[..., None] * dir_l - rv
    direction = np.where(circle[..., None], dir_c, dir_l)
    u = np.where(circle[..., None], u_c, u_l)
    if with_dist:
        return vel[:, None, :] + 0.5 * u, direction, dist_sq
    return vel[:, None, :] + 0.5 * u, direction


# ------------------------------------------------------------------------------------------------ linear programs
def _lp1(P, D, active, i: int, radius, opt, direction_opt: bool, mask):
    """linearProgram1 on constraint line i for the rows in ``mask``: the point of line i, inside the speed disc and
    all earlier (active) lines, that is optimal.  Returns (ok [B], result [B, 2])."""
    p, d = P[:, i], D[:, i]
    dp = _dot(p, d)
    disc = dp * dp + radius * radius - _dot(p, p)
    ok = mask & (disc >= 0.0)
    sq = np.sqrt(np.where(disc >= 0.0, disc, 0.0))
    t_left, t_right = -dp - sq, -dp + sq
    for j in range(i):
        pj, dj = P[:, j], D[:, j]
        den = _det(d, dj)
        num = _det(dj, p - pj)
        live = ok & active[:, j]
        par = np.abs(den) <= RVO_EPSILON
        ok = ok & ~(live & par & (num < 0.0))
        upd = live & ~par & ok
        with np.errstate(invalid="ignore", divide="ignore"):
            t = num / den
        t_right = np.where(upd & (den >= 0.0), np.minimum(t_right, t), t_right)
        t_left = np.where(upd & (den < 0.0), np.maximum(t_left, t), t_left)
        ok = ok & ~(upd & (t_left > t_right))
    if direction_opt:
        t = np.where(_dot(opt, d) > 0.0, t_right, t_left)
    else:
        t = _dot(d, opt - p)
        t = np.where(t < t_left, t_left, np.where(t > t_right, t_right, t))
    return ok, p + t[:, None] * d


def _lp2(P, D, active, radius, opt, direction_opt: bool, mask):
    """linearProgram2 for the rows in ``mask``.  Returns (failed [B], fail_index [B], result [B, 2]); rows outside the
    mask come back with failed = False and an unspecified result."""
    L = P.shape[1]
    if direction_opt:
        result = opt * radius[:, None]
    else:
        n2 = _dot(opt, opt)
        with np.errstate(invalid="ignore", divide="ignore"):
            scaled = opt / np.sqrt(n2)[:, None] * radius[:, None]
        result = np.where((n2 > radius * radius)[:, None], scaled, opt)
    alive = mask.copy()
    failed = np.zeros_like(mask)
    fail_idx = np.full(mask.shape, L, dtype=np.int64)
    for i in range(L):
        viol = alive & active[:, i] & (_det(D[:, i], P[:, i] - result) > 0.0)
        if not viol.any():
            continue
        ok, r = _lp1(P, D, active, i, radius, opt, direction_opt, viol)
        result = np.where((viol & ok)[:, None], r, result)
        bad = viol & ~ok
        failed |= bad
        fail_idx = np.where(bad, i, fail_idx)
        alive &= ~bad
    return failed, fail_idx, result


def _lp3(P, D, begin, radius, result, mask, is_obst=None):
    """linearProgram3: for rows whose program is infeasible, the velocity that violates the half-planes least.  ``is_obst`` [B, L]
    marks obstacle half-planes (crowd_env.orca_plus_velocities): they enter every projected program unchanged - hard constraints,
    as RVO2 keeps its first ``numObstLines`` lines."""
    B, L, _ = P.shape
    distance = np.zeros(B)
    idx = np.arange(L)
    for i in range(L):
        pi, di = P[:, i], D[:, i]
        cond = mask & (begin <= i) & (_det(di, pi - result) > distance)
        if not cond.any():
            continue
        determinant = _det(di[:, None, :], D)                                   # [B, L]
        par = np.abs(determinant) <= RVO_EPSILON
        same = _dot(di[:, None, :], D) > 0.0
        act = (idx[None, :] < i) & ~(par & same)
        with np.errstate(invalid="ignore", divide="ignore"):
            t = _det(D, pi[:, None, :] - P) / determinant
            point = np.where(par[..., None], 0.5 * (pi[:, None, :] + P), pi[:, None, :] + t[..., None] * di[:, None, :])
            dd = D - di[:, None, :]
            dd = dd / np.sqrt(_dot(dd, dd))[..., None]
        if is_obst is not None:
            act = np.where(is_obst, True, act & ~is_obst)
            point = np.where(is_obst[..., None], P, point)
            dd = np.where(is_obst[..., None], D, dd)
        point = np.where(act[..., None], point, 0.0)
        dd = np.where(act[..., None], dd, 0.0)
        opt = np.stack([-di[:, 1], di[:, 0]], axis=-1)
        failed, _, r = _lp2(point, dd, act, radius, opt, True, cond)
        result = np.where((cond & ~failed)[:, None], r, result)
        distance = np.where(cond, _det(di, pi - result), distance)
    return result


def orca_velocities(pos, vel, radius, pref, max_speed, time_horizon: float = 2.0, time_step: float = 0.25,
                    neighbor_dist: float = 10.0, max_neighbors: int = 10):
    """New ORCA velocity of EVERY agent of every episode, each as the ego of its own program (``orca.py:96-131``).

    pos, vel, pref [E, n, 2]; radius, max_speed [E, n] (the radius already holds the + 0.01 + safety space the reference
    adds).  Only the ``max_neighbors`` nearest agents closer than ``neighbor_dist`` constrain an agent, as in RVO2
    (``Agent::insertAgentNeighbor``; the reference passes 10 and 10 m, ``orca.py:62-63, 94``).  Returns [E, n, 2].
    n = 1 returns the preferred velocity clipped to the speed limit."""
    E, n, _ = pos.shape
    B = E * n
    others = np.array([[j for j in range(n) if j != i] for i in range(n)], dtype=np.int64).reshape(n, max(n - 1, 0))
    ego = lambda a: a.reshape((B,) + a.shape[2:])
    oth = lambda a: a[:, others].reshape((B, n - 1) + a.shape[2:])
    P, D, dsq = orca_lines(ego(pos), ego(vel), ego(radius), oth(pos), oth(vel), oth(radius), time_horizon, time_step,
                           with_dist=True)
    active = (dsq < neighbor_dist * neighbor_dist) & (np.arange(n - 1)[None, :] < max_neighbors)
    allrows = np.ones(B, dtype=bool)
    if not active.all():       # neighbours left out: their half-planes must not constrain (the programs only look at active lines)
        far = ~active
        P = np.where(far[..., None], 0.0, P)
        D = np.where(far[..., None], np.array([1.0, 0.0]), D)     # a line through the origin along x ...
        P = np.where(far[..., None], np.array([0.0, -1e9]), P)    # ... moved far below: every velocity lies on its free side
    failed, fail_idx, result = _lp2(P, D, active, ego(max_speed), ego(pref), False, allrows)
    if failed.any():
        result = _lp3(P, D, fail_idx, ego(max_speed), result, failed)
    return result.reshape(E, n, 2)


# ------------------------------------------------------------------------------------------------ scenario + rollout
@dataclass
class CrowdConfig:
    """The fields of ``configs/env.config`` / ``orca.py`` the circle-crossing simulation reads."""
    time_step: float = 0.25                 # [env] time_step
    circle_radius: float = 4.0              # [sim] circle_radius (CrowdNav's circle crossing; the shipped 1.0 is the hallway's)
    square_width: float = 5.0               # [sim] square_width (square crossing; the shipped value)
    human_radius: float = 0.20              # [humans] radius
    human_v_pref: float = 1.5               # [humans] v_pref (drawn from U(0.5, 1.5) with randomize_attributes)
    robot_radius: float = 0.25              # [robot] radius
    robot_v_pref: float = 1.0               # [robot] v_pref
    randomize_attributes: bool = True       # [env] randomize_attributes
    discomfort_dist: float = 0.2            # [reward] discomfort_dist
    safety_space: float = 0.0               # orca.py:61
    time_horizon: float = 2.0               # orca.py:64
    time_horizon_obst: float = 0.5          # orca.py:65 (no obstacles here: unused, handed to rvo2 all the same)
    neighbor_dist: float = 10.0             # orca.py:62
    max_neighbors: int = 10                 # orca.py:63
    orca_default_radius: float = 0.3        # orca.py:66-67: the simulator's defaults for new agents (every agent overrides them)
    orca_default_max_speed: float = 1.0
    robot_visible: bool = True              # [robot] visible


def place_circle_crossing_humans(N: int, rng: np.random.Generator, cfg: CrowdConfig):
    """``generate_circle_crossing_human`` x N for ONE episode (``crowd_sim_plus.py:441-443, 454-481``), with the reference's
    draw order from ``rng``: per human the preferred speed U(0.5, 1.5) when attributes are randomised, then per attempt the
    angle and the two noise terms; goal at the antipode; re-drawn while closer than radius + radius + discomfort distance to
    an earlier agent's start or goal (the robot at (0, -R) -> (0, R) is agent 0).  -> pos, goal [N, 2], v_pref [N].
    Bit-equal to the reference's lines on the same generator (``tests/golden/episodes_placement_*.npz``)."""
    R = cfg.circle_radius
    starts, goals, radii = [(0.0, -R)], [(0.0, R)], [cfg.robot_radius]
    pos, goal, vp = np.zeros((N, 2)), np.zeros((N, 2)), np.full(N, cfg.human_v_pref)
    for h in range(N):
        if cfg.randomize_attributes:
            vp[h] = rng.uniform(0.5, 1.5)
        for _ in range(100000):
            angle = rng.random() * np.pi * 2
            px_noise = (rng.random() - 0.5) * vp[h]
            py_noise = (rng.random() - 0.5) * vp[h]
            px = R * np.cos(angle) + px_noise
            py = R * np.sin(angle) + py_noise
            collide = False
            for (sx, sy), (gx, gy), r in zip(starts, goals, radii):
                min_dist = cfg.human_radius + r + cfg.discomfort_dist
                if np.linalg.norm((px - sx, py - sy)) < min_dist or np.linalg.norm((px - gx, py - gy)) < min_dist:
                    collide = True
                    break
            if not collide:
                break
        else:
            raise RuntimeError("circle crossing placement did not converge (circle too small for the crowd?)")
        pos[h], goal[h] = (px, py), (-px, -py)
        starts.append((px, py))
        goals.append((-px, -py))
        radii.append(cfg.human_radius)
    return pos, goal, vp


def place_square_crossing_humans(N: int, rng: np.random.Generator, cfg: CrowdConfig):
    """``generate_square_crossing_human`` x N for ONE episode (``crowd_sim_plus.py:436-439, 484-520``), with the reference's draw
    order: per human the preferred speed U(0.5, 1.5) when attributes are randomised, one draw for the side of the y axis it
    starts on, then start positions (x on its side of the square, y anywhere in it) until one is at least radius + radius +
    discomfort distance from every earlier agent's START, then goal positions on the other side until one is that far from
    every earlier agent's GOAL (the robot at (0, -R) -> (0, R) is agent 0).  -> pos, goal [N, 2], v_pref [N].
    Bit-equal to the reference's lines on the same generator (``tests/golden/episodes_square_placement_*.npz``)."""
    R, W = cfg.circle_radius, cfg.square_width
    starts, goals, radii = [(0.0, -R)], [(0.0, R)], [cfg.robot_radius]
    pos, goal, vp = np.zeros((N, 2)), np.zeros((N, 2)), np.full(N, cfg.human_v_pref)

    def draw(side, taken):
        for _ in range(100000):
            x = rng.random() * W * 0.5 * side
            y = (rng.random() - 0.5) * W
            if not any(np.linalg.norm((x - tx, y - ty)) < cfg.human_radius + r + cfg.discomfort_dist for (tx, ty), r in zip(taken, radii)):
                return x, y
        raise RuntimeError("square crossing placement did not converge (square too small for the crowd?)")

    for h in range(N):
        if cfg.randomize_attributes:
            vp[h] = rng.uniform(0.5, 1.5)
        sign = -1 if rng.random() > 0.5 else 1
        pos[h] = draw(sign, starts)
        goal[h] = draw(-sign, goals)
        starts.append(tuple(pos[h]))
        goals.append(tuple(goal[h]))
        radii.append(cfg.human_radius)
    return pos, goal, vp


def episode_rng(seed: int, episode: int) -> np.random.Generator:
    """The generator of one episode: a child stream of ``seed``, so that an episode's crowd does not depend on how many
    episodes are generated with it, and different seeds (ranks of a sweep) never share a stream."""
    return np.random.default_rng(np.random.SeedSequence(entropy=int(seed), spawn_key=(int(episode),)))


PLACEMENT = {"circle_crossing": place_circle_crossing_humans, "square_crossing": place_square_crossing_humans}


def crossing_starts(E: int, N: int, seed: int, cfg: CrowdConfig, rule: str = "circle_crossing") -> Dict[str, np.ndarray]:
    """Start record of E crossing episodes (``rule``: ``circle_crossing`` or ``square_crossing``, crowd_sim_plus.py:436-443): every
    episode placed by the rule's function on its own generator ``episode_rng(seed, e)``.  Index 0 of the agent axis is the
    robot, at (0, -circle_radius) -> (0, circle_radius) under either rule (crowd_sim_plus.py:661)."""
    if rule not in PLACEMENT:
        raise ValueError(f"unknown crossing rule {rule!r} (the hallway rules live in crowd_env.py)")
    pos = np.zeros((E, N + 1, 2))
    goal = np.zeros((E, N + 1, 2))
    rad = np.full((E, N + 1), cfg.human_radius)
    vp = np.full((E, N + 1), cfg.human_v_pref)
    pos[:, 0] = (0.0, -cfg.circle_radius)
    goal[:, 0] = (0.0, cfg.circle_radius)
    rad[:, 0], vp[:, 0] = cfg.robot_radius, cfg.robot_v_pref
    for e in range(E):
        pos[e, 1:], goal[e, 1:], vp[e, 1:] = PLACEMENT[rule](N, episode_rng(seed, e), cfg)
    return dict(pos=pos, goal=goal, radius=rad, v_pref=vp)


def circle_crossing_starts(E: int, N: int, seed: int, cfg: CrowdConfig) -> Dict[str, np.ndarray]:
    return crossing_starts(E, N, seed, cfg, "circle_crossing")


def orca_call_parameters(cfg: CrowdConfig, pos, vel, goal, radius, v_pref) -> Dict[str, np.ndarray]:
    """What the reference hands to rvo2 for ONE agent's step (``orca.py:93-129``), for every agent of every episode as
    the ego: arrays [E, n, ...] of the inflated radius, the speed limit and the preferred velocity, plus the scalars.
    Pinned by ``tests/golden/episodes_orca_calls_*.npz`` (recorded from the reference's own lines)."""
    to_goal = goal - pos
    # np.linalg.norm of a 1-D vector (what orca.py:114 calls) is sqrt(dot(x, x)) with the BLAS dot's fused multiply-add;
    # the batched matmul goes through the same BLAS and gives the same bits (the axis=-1 norm differs in the last place)
    speed = np.sqrt(to_goal[..., None, :] @ to_goal[..., :, None])[..., 0]
    with np.errstate(invalid="ignore", divide="ignore"):
        pref = np.where(speed > 1.0, to_goal / speed, to_goal)          # orca.py:113-115: unit length only beyond 1 m
    return dict(radius=radius + 0.01 + cfg.safety_space,                # orca.py:104-108
                max_speed=v_pref,                                       # the ego's limit is its v_pref (orca.py:105)
                pref=pref, neighbor_dist=cfg.neighbor_dist, max_neighbors=cfg.max_neighbors,
                time_horizon=cfg.time_horizon, time_horizon_obst=cfg.time_horizon_obst, time_step=cfg.time_step,
                default_radius=cfg.orca_default_radius, default_max_speed=cfg.orca_default_max_speed)


def simulate_circle_crossing(E: int, N: int, steps: int, seed: int, cfg: Optional[CrowdConfig] = None
                             ) -> Dict[str, np.ndarray]:
    return simulate_crossing(E, N, steps, seed, cfg, "circle_crossing")


def simulate_crossing(E: int, N: int, steps: int, seed: int, cfg: Optional[CrowdConfig] = None, rule: str = "circle_crossing"
                      ) -> Dict[str, np.ndarray]:
    """E independent crossing episodes (``rule``: circle or square crossing) with N ORCA humans and an ORCA robot, ``steps``
    simulator steps.

    Returns human_xy [E, steps + 1, N, 2], robot_xy [E, steps + 1, 2], human_vel [E, steps + 1, N, 2], stamps
    [steps + 1] and the start record (goals, radii, v_pref): what ``update_state_hists`` is fed step by step in the
    reference's loop, for all episodes at once."""
    cfg = cfg or CrowdConfig()
    st = crossing_starts(E, N, seed, cfg, rule)
    pos, goal = st["pos"].copy(), st["goal"]
    vel = np.zeros_like(pos)
    traj = np.zeros((E, steps + 1, N + 1, 2))
    vels = np.zeros((E, steps + 1, N + 1, 2))
    traj[:, 0] = pos
    sl = slice(None) if cfg.robot_visible else slice(1, None)
    for s in range(steps):
        par = orca_call_parameters(cfg, pos, vel, goal, st["radius"], st["v_pref"])
        pref = par["pref"]
        new_vel = vel.copy()
        new_vel[:, sl] = orca_velocities(pos[:, sl], vel[:, sl], par["radius"][:, sl], pref[:, sl], par["max_speed"][:, sl],
                                         par["time_horizon"], par["time_step"], par["neighbor_dist"], par["max_neighbors"])
        if not cfg.robot_visible:       # humans do not see the robot; it still heads for its goal
            new_vel[:, 0] = pref[:, 0] * np.minimum(1.0, st["v_pref"][:, 0])[:, None]
        vel = new_vel
        pos = pos + vel * cfg.time_step                             # holonomic step (agent.py compute_position)
        traj[:, s + 1] = pos
        vels[:, s + 1] = vel
    return dict(human_xy=traj[:, :, 1:], robot_xy=traj[:, :, 0], human_vel=vels[:, :, 1:],
                stamps=np.arange(steps + 1) * cfg.time_step, goal=goal, radius=st["radius"], v_pref=st["v_pref"])


def history_windows(sim: Dict[str, np.ndarray], frame: int, num_hist_frames: int = 6) -> Tuple[np.ndarray, np.ndarray]:
    """The last ``num_hist_frames`` frames ending at simulator step ``frame``: human_xy [E, F, N, 2], robot_xy [E, F, 2]
    - the input of ``scene.build_scenes_batched`` / ``forecaster.predict_batch``."""
    lo = frame - num_hist_frames + 1
    if lo < 0:
        raise ValueError("not enough simulated frames for a full history window")
    return sim["human_xy"][:, lo:frame + 1], sim["robot_xy"][:, lo:frame + 1]
