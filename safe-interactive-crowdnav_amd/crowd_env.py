"""CrowdSimPlus semantics around the batched episode generator: the reference's scenarios with static
obstacles, its wall-constrained actions and what one ``step()`` decides - collision, frozen robot, goal, timeout, rewards.

The reference steps one ``CrowdSimPlus`` episode at a time (``crowd_sim_plus/envs/crowd_sim_plus.py``).  Its shipped
configuration (``sicnav_diffusion/configs/env.config:15-23``) is NOT the circle crossing of ``episodes.py`` but the ``hallway``
family: two walls 1.75 m apart, optionally a door between them, humans that walk up or down the corridor, ORCA humans that
see the walls (``policy/orca_plus.py``).  This module restates, for E episodes at once and in NumPy on the host (where the
simulator and the MPC live):

  ``static_obstacles``        ``generate_static_obstacles``        ``crowd_sim_plus.py:322-421``   wall / door segments per rule
  ``door_subgoal``            ``Human.get_g_xy``                    ``utils/human_plus.py:19-52``    intermediate goal in front of a door
  ``place_hallway_humans``    ``generate_hallway_human``            ``crowd_sim_plus.py:522-607``   placement, with the draw order
  ``constrain_actions``       ``constrain_agent_action_exact``      ``crowd_sim_plus.py:869-989``   an action cut short at a wall
  ``constrain_unicycle_actions``  the same, ``ActionRot`` branch     ``crowd_sim_plus.py:976-987``   the MPC's unicycle: |v| shrinks, r stays
  ``step_outcomes``           the outcome block of ``step()``       ``crowd_sim_plus.py:1067-1166`` flags, reward terms, done
  ``sfm_velocities``          ``SFM.predict``                       ``policy/social_force.py:38-95`` social-force humans
  ``orca_plus_parameters``    what ``ORCAPlus.predict`` hands rvo2  ``policy/orca_plus.py:43-84``
  ``obstacle_orca_lines``     RVO2's obstacle half-planes           RVO2 Library 2.0.2 ``Agent::computeNewVelocity``
  ``simulate_hallway``        ``reset`` + ``step`` x n              ``crowd_sim_plus.py:609-721, 1025-1258``   E episodes at once

PARITY.  Everything in the first eight rows is plain Python / NumPy in the reference and is pinned by fixtures that
``tests/golden/make_golden_env.py`` generates by executing the reference's own lines against stand-in objects
(``tests/golden/env_*.npz``; ``tests/test_crowd_env.py``).  With the social-force humans (``human_policy="sfm"``) an episode holds
nothing else: whole episodes of ``simulate_hallway`` - placement, observations, forces, wall constraint, door sub-goals, outcome
block, position updates - land on the positions the reference's lines produce step by step (``env_rollout_sfm_*.npz``, 40-50
steps, <= 1e-10 m).  With the shipped ``orca_plus`` humans the velocities come out of rvo2, an un-vendored C++ dependency that is
absent from the reference tree and from this image: ``obstacle_orca_lines`` (like the agent-agent half-planes and the linear
programs of ``episodes.py``) restates its published algorithm and is UNPINNED, checked against a scalar restatement kept with the
test infrastructure, by brute force and through properties (no agent of a generated crowd ever enters a wall).  The humans are
holonomic (``ActionXY``); the robot is either that or the MPC's unicycle (``ActionRot``, ``sicnav_acados.py:143``): wall constraint,
outcome block, position / heading update in their ``ActionRot`` branches, pinned the same way incl. two whole episodes
(``env_rotconstrain_*.npz``, ``env_step_outcomes_unicycle_*.npz``, ``env_rollout_sfm_unicycle_*.npz``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

DOOR_RULES = ("hallway_static", "hallway_static_with_back", "hallway_bottleneck", "hallway_squeeze")
SUBGOAL_RULES = ("hallway_static", "hallway_static_with_back", "hallway_bottleneck")      # human_plus.py:30
HALLWAY_RULES = ("hallway", "hallway_static", "hallway_bottleneck", "hallway_squeeze", "rectangle", "hallway_static_with_back",
                 "left_wall", "no_walls")                                                 # crowd_sim_plus.py:445


@dataclass(frozen=True)
class Geometry:
    """``[sim]`` of env.config (the shipped values) + the robot radius the door is sized by."""
    circle_radius: float = 1.0
    rect_width: float = 1.75
    rect_height: float = 4.0
    robot_radius: float = 0.25


@dataclass(frozen=True)
class Doors:
    """Door geometry of the rules that have one (``crowd_sim_plus.py:333-346``)."""
    y_max: float
    y_min: float
    x_mid: float
    y_mid_max: float
    y_mid_min: float
    width: float


def static_obstacles(rule: str, geo: Geometry = Geometry()) -> Tuple[np.ndarray, Optional[Doors]]:
    """The wall / door line segments of a scenario rule as [L, 2, 2] (segment, end, xy) and its door geometry (None for the
    rules without a door).  ``no_walls`` and the crossing rules have no segments."""
    R, W, H, rr = geo.circle_radius, geo.rect_width, geo.rect_height, geo.robot_radius
    segs, doors = [], None
    if rule in DOOR_RULES:
        y_max = R - rr * 2.0
        y_min = -R + rr * 2.0
        x_mid = 0.0
        y_mid_max = y_max + (y_min - y_max) * 0.40
        y_mid_min = y_max + (y_min - y_max) * (1.0 - 0.40)
        width = 0.5 * W if rule == "hallway_squeeze" else 1.0
        doors = Doors(y_max, y_min, x_mid, y_mid_max, y_mid_min, width)
        xl = x_mid - width / 2.0
        xl_mid = xl + ((-W * 0.5) - xl) * 0.75
        xr = x_mid + width / 2.0
        xr_mid = xr + (W * 0.5 - xr) * 0.75
        if rule == "hallway_squeeze":
            segs = [[(-W * 0.5, -R * 2.5), (xl, 0)], [(xl, 0), (-W * 0.5, R * 2.5)],
                    [(W * 0.5, -R * 2.5), (xr, 0)], [(xr, 0), (W * 0.5, R * 2.5)]]
        else:
            segs = [[(-W * 0.5, -H), (-W * 0.5, H)], [(W * 0.5, -H), (W * 0.5, H)]]
            if "hallway_static" in rule:
                segs += [[(-W * 0.5, y_min), (xl_mid, y_min)], [(xl_mid, y_min), (xl, y_mid_min)], [(xl, y_mid_min), (xl, y_mid_max)],
                         [(xl, y_mid_max), (xl_mid, y_max)], [(xl_mid, y_max), (-W * 0.5, y_max)],
                         [(W * 0.5, y_min), (xr_mid, y_min)], [(xr_mid, y_min), (xr, y_mid_min)], [(xr, y_mid_min), (xr, y_mid_max)],
                         [(xr, y_mid_max), (xr_mid, y_max)], [(xr_mid, y_max), (W * 0.5, y_max)]]
            else:      # hallway_bottleneck
                segs += [[(-W * 0.5, 0), (xl, 0)], [(xr, 0), (W * 0.5, 0)]]
            if rule == "hallway_static_with_back":
                segs += [[(-W * 0.5, -H * 0.5), (W * 0.5, -H * 0.5)], [(-W * 0.5, H * 0.5), (W * 0.5, H * 0.5)]]
    elif rule == "hallway":
        segs = [[(-W * 0.5, -H), (-W * 0.5, H)], [(W * 0.5, -H), (W * 0.5, H)]]
    elif rule == "rectangle":
        segs = [[(-W * 0.5, -H * 0.5), (-W * 0.5, H * 0.5)], [(W * 0.5, -H * 0.5), (W * 0.5, H * 0.5)],
                [(-W * 0.5, -H * 0.5), (W * 0.5, -H * 0.5)], [(-W * 0.5, H * 0.5), (W * 0.5, H * 0.5)]]
    elif rule == "left_wall":
        segs = [[(-W * 0.5, -H * 1000), (-W * 0.5, H * 1000)]]
    elif rule not in ("no_walls", "circle_crossing", "square_crossing"):
        raise ValueError(f"unknown scenario rule {rule!r}")
    return np.array(segs, dtype=np.float64).reshape(-1, 2, 2), doors


def door_subgoal(pos, final_goal, rule: str, doors: Optional[Doors], n_obstacles: int = 1):
    """``Human.get_g_xy``: the goal a human currently walks to.  In the rules with a door between start and goal
    (``hallway_static*``, ``hallway_bottleneck``) a human whose way leads through the door heads for the door's centre until it
    is within half a door width of it.  pos, final_goal [..., 2] -> [..., 2]."""
    pos, final_goal = np.asarray(pos, np.float64), np.asarray(final_goal, np.float64)
    if doors is None or rule not in SUBGOAL_RULES or n_obstacles == 0:
        return np.broadcast_to(final_goal, np.broadcast_shapes(pos.shape, final_goal.shape)).copy()
    lo = np.minimum(pos[..., 1], final_goal[..., 1])
    hi = np.maximum(pos[..., 1], final_goal[..., 1])
    through = (lo < doors.y_mid_min) & (hi > doors.y_mid_max)
    ig = np.array([doors.x_mid, 0.5 * (doors.y_min + doors.y_max)])
    d = ig - pos
    near = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) <= doors.width / 2.0
    return np.where((through & ~near)[..., None], ig, final_goal)


def _norm(v):
    """Euclidean length along the last axis with the bits of ``np.linalg.norm`` on a 1-D vector (sqrt of the BLAS dot, which fuses
    its multiply-adds): the batched matmul goes through the same BLAS."""
    v = np.asarray(v, np.float64)
    return np.sqrt(v[..., None, :] @ v[..., :, None])[..., 0, 0]


def _point_segment_dist(x1, y1, x2, y2, x3, y3):
    """``point_to_segment_dist`` (utils_plus.py:73-96), arrays welcome."""
    px, py = x2 - x1, y2 - y1
    den = px * px + py * py
    with np.errstate(invalid="ignore", divide="ignore"):
        u = np.where(den > 0, ((x3 - x1) * px + (y3 - y1) * py) / den, 0.0)
    u = np.clip(u, 0.0, 1.0)
    dx, dy = x1 + u * px - x3, y1 + u * py - y3
    return np.sqrt(dx * dx + dy * dy)


def place_hallway_humans(N: int, rng: np.random.Generator, rule: str, geo: Geometry = Geometry(), human_radius: float = 0.20,
                         human_v_pref: float = 1.5, randomize_attributes: bool = True, discomfort_dist: float = 0.2
                         ) -> Dict[str, np.ndarray]:
    """``generate_hallway_human`` x N for ONE episode (``crowd_sim_plus.py:444-447, 522-607``) with the reference's draw order:
    per attempt the preferred speed (when attributes are randomised - re-drawn on every attempt), four coins (direction of
    travel, side of the corridor, crossing probability, crossing) and the two start coordinates; then the two goal coordinates.
    A start is rejected within the discomfort distance of the robot, on top of an earlier agent or within radius + 1 cm of a
    wall; a goal on top of an earlier agent's CURRENT goal (its door sub-goal where one applies) or within a radius of a wall;
    every rejection widens the band the y coordinates are drawn from by 10 %.  The robot stands at (0, -R) and goes to (0, R).
    -> pos, final_goal, goal (after the door rule), v_pref, theta.  Bit-equal to the reference's lines on the same generator
    (``tests/golden/env_hallway_placement_*.npz``)."""
    segs, doors = static_obstacles(rule, geo)
    R, W = geo.circle_radius, geo.rect_width
    robot = dict(px=0.0, py=-R, gx=0.0, gy=R, radius=geo.robot_radius)
    placed = []
    out = dict(pos=np.zeros((N, 2)), final_goal=np.zeros((N, 2)), goal=np.zeros((N, 2)), v_pref=np.full(N, human_v_pref),
               theta=np.zeros(N))
    norm2 = lambda x, y: float(np.linalg.norm((x, y)))
    for h in range(N):
        v_pref = human_v_pref
        eff_h = geo.rect_height
        for _ in range(100000):
            if randomize_attributes:
                v_pref = rng.uniform(0.5, 1.5)
            dir_sign = 1 if rng.random() < 0.15 else -1
            right_num = 0.8 if dir_sign > 0 else 1 - 0.8
            wor_sign = -1 if rng.random() < right_num else 1
            prob_cross = 0.3
            if rng.random() < right_num:
                prob_cross = 1 - prob_cross
            cross_sign = -wor_sign if rng.random() < prob_cross else wor_sign
            px = (rng.random()) * 0.5 * wor_sign * (W - human_radius * 2)
            py = (rng.random()) * 0.25 * dir_sign * R * (eff_h - human_radius * 2)
            collide = norm2(px - robot["px"], py - robot["py"]) < human_radius + robot["radius"] + discomfort_dist
            if not collide:
                for a in [robot] + placed:
                    if norm2(px - a["px"], py - a["py"]) < human_radius + a["radius"]:
                        collide = True
                        break
            if not collide:
                for s in segs:
                    if abs(float(_point_segment_dist(s[0, 0], s[0, 1], s[1, 0], s[1, 1], px, py))) < human_radius + 0.01:
                        collide = True
                        break
            if collide:
                eff_h *= 1.1
                continue
            gx = (rng.random()) * 0.5 * cross_sign * (W - human_radius * 2)
            gy = (rng.random()) * 0.5 * -dir_sign * R * (eff_h - human_radius * 2)
            collide = False
            for a in [robot] + placed:
                if norm2(gx - a["gx"], gy - a["gy"]) < human_radius + a["radius"]:
                    collide = True
                    break
            if not collide:
                for s in segs:
                    if abs(float(_point_segment_dist(s[0, 0], s[0, 1], s[1, 0], s[1, 1], gx, gy))) < human_radius:
                        collide = True
                        break
            if not collide:
                break
            eff_h *= 1.1
        else:
            raise RuntimeError("hallway placement did not converge")
        cur = door_subgoal(np.array([px, py]), np.array([gx, gy]), rule, doors, len(segs))
        placed.append(dict(px=px, py=py, gx=float(cur[0]), gy=float(cur[1]), radius=human_radius))
        out["pos"][h], out["final_goal"][h], out["goal"][h] = (px, py), (gx, gy), cur
        out["v_pref"][h], out["theta"][h] = v_pref, np.arctan2(gy - py, gx - px)
    return out


# ------------------------------------------------------------------------------------------------ wall-constrained actions
def _closest_between_segments(a0, a1, b0, b1):
    """``closest_distance_between_line_segments`` (utils_plus.py:205-340) in the plane: the closest points pA on segment a (the
    wall) and pB on segment b (the agent's path) and their distance, with the reference's conventions for parallel overlaps."""
    A, B = a1 - a0, b1 - b0
    magA, magB = float(np.sqrt(A @ A)), float(np.sqrt(B @ B))
    if magA < 1e-8:
        a1, A, _A = a0, np.zeros(2), np.zeros(2)
    else:
        _A = A / magA
    if magB < 1e-8:
        b1, B, _B = b0, np.zeros(2), np.zeros(2)
    else:
        _B = B / magB
    cross = _A[0] * _B[1] - _A[1] * _B[0]
    denom = cross * cross
    dist = lambda p, q: float(np.sqrt((p - q) @ (p - q)))
    if not denom:
        d0, d1 = float(_A @ (b0 - a0)), float(_A @ (b1 - a0))
        if d0 <= 0 >= d1:
            return (a0, b0, dist(a0, b0)) if abs(d0) < abs(d1) else (a0, b1, dist(a0, b1))
        if d0 >= magA <= d1:
            return (a1, b0, dist(a1, b0)) if abs(d0) < abs(d1) else (a1, b1, dist(a1, b1))
        if dist(_A, _B) < 1e-8 or magB < 1e-8:
            a0f, _Af = a0, _A
        else:
            a0f, _Af = a1, -_A
        d0f = float(_Af @ (b0 - a0f))
        if d0f >= 0:
            pB = b0
            pA = a0f + _Af * float(_Af @ (pB - a0f))
        else:
            pA = a0f
            pB = b0 + _B * float(_B @ (pA - b0))
        return pA, pB, dist(pA, pB)
    t = b0 - a0
    t0 = cross * (t[0] * _B[1] - t[1] * _B[0]) / denom
    t1 = cross * (t[0] * _A[1] - t[1] * _A[0]) / denom
    pA, pB = a0 + _A * t0, b0 + _B * t1
    if t0 < 0:
        pA = a0
    elif t0 > magA:
        pA = a1
    if t1 < 0:
        pB = b0
    elif t1 > magB:
        pB = b1
    if t0 < 0 or t0 > magA:
        pB = b0 + _B * min(max(float(_B @ (pA - b0)), 0.0), magB)
    if t1 < 0 or t1 > magB:
        pA = a0 + _A * min(max(float(_A @ (pB - a0)), 0.0), magA)
    return pA, pB, dist(pA, pB)


def _constrain_one(cur, act, r, dt, segs, rot=None):
    """``constrain_agent_action_exact`` for one agent against the segments that its step comes within r of.  Holonomic: ``act`` is
    the velocity, the constrained velocity comes back.  ``rot`` = (theta, v, r_step) - the unicycle of ``agent_plus.py:175-185``
    (heading theta + r_step, signed speed v): the constrained signed speed comes back (``crowd_sim_plus.py:976-987``)."""
    if rot is not None:
        th = rot[0] + rot[2]
        fut = np.array([cur[0] + np.cos(th) * rot[1] * dt, cur[1] + np.sin(th) * rot[1] * dt])
    else:
        fut = cur + act * dt
    move = fut - cur
    move_mag = float(np.sqrt(move @ move))
    hits = []
    for s in segs:
        pA, pB, cd = _closest_between_segments(s[0], s[1], cur, fut)
        if cd - r < 0.0:
            hits.append((s, cd, pA, pB))
    final = act.copy() if rot is None else None
    final_v = rot[1] if rot is not None else None
    nrm = lambda v: float(np.sqrt(v @ v))
    for s, cd, pA, pB in hits:
        if (nrm(pA - s[0]) < 1e-8 or nrm(pA - s[1]) < 1e-8) and nrm(pA - pB) > 1e-8:
            # against an END of the wall: stop where the agent's disc touches the end point
            dvec = pB - cur
            dmag = nrm(dvec)
            if dmag > 0.0 and nrm(pA - cur) - r < 1e-4 and float(move @ (pA - cur)) > -1e-8:
                u, redux = dvec / dmag, dmag                     # touching already and not moving away: stay
            elif dmag > 0.0:
                u = dvec / dmag
                alpha = np.arccos(np.clip(float(-dvec @ (pA - pB)) / (dmag * cd), -1.0, 1.0))
                if alpha == np.pi:
                    redux = r - cd
                else:
                    gamma = np.arcsin(cd * np.sin(alpha) / r)
                    beta = np.pi - alpha - gamma
                    redux = r * np.sin(beta) / np.sin(alpha) + 1e-7
            else:
                u, redux = dvec, 0.0
            fin = cur + u * max(dmag - redux, 0)
        else:
            # against the wall's line, as if it went on for ever
            px, py = s[1, 0] - s[0, 0], s[1, 1] - s[0, 1]
            uu = ((cur[0] - s[0, 0]) * px + (cur[1] - s[0, 1]) * py) / (px * px + py * py)
            cl = np.array([s[0, 0] + uu * px, s[0, 1] + uu * py])
            if move_mag > 0.0 and nrm(cl - cur) - r < 1e-4 and float(move @ (cl - cur)) > -1e-8:
                fin = cur
            elif move_mag > 0.0:
                x1, y1, x2, y2 = s[0, 0], s[0, 1], s[1, 0], s[1, 1]
                x3, y3, x4, y4 = cur[0], cur[1], cur[0] + move[0], cur[1] + move[1]
                den = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4)
                ix = ((x1 * y2 - y1 * x2) * (x3 - x4) - (x1 - x2) * (x3 * y4 - y3 * x4)) / den
                iy = ((x1 * y2 - y1 * x2) * (y3 - y4) - (y1 - y2) * (x3 * y4 - y3 * x4)) / den
                dc0 = np.sqrt((cur[0] - cl[0]) ** 2 + (cur[1] - cl[1]) ** 2)
                fin = cur + np.array([ix - cur[0], iy - cur[1]]) * max(0.0, (dc0 - (r + 1e-7)) / dc0)
            else:
                fin = cur
        if rot is not None:                                                     # the rotation stays, |v| shrinks
            step = fin - cur
            vv = float(np.sqrt(step @ step)) / dt
            if rot[1] > 0:
                if vv < final_v:
                    final_v = vv
            elif -vv > final_v:
                final_v = -vv
            continue
        cand = (fin - cur) / dt
        if cand[0] ** 2 + cand[1] ** 2 < final[0] ** 2 + final[1] ** 2:       # the slowest of the candidates wins
            final = cand
    return final if rot is None else final_v


def constrain_actions(pos, action, radius, time_step: float, segments) -> np.ndarray:
    """``constrain_agent_action_exact`` for B holonomic agents: pos, action [B, 2] (velocities), radius [B] or scalar ->
    the velocities actually taken.  An action whose straight step keeps the agent's disc clear of every segment comes back
    unchanged (bitwise); otherwise the step is cut short where the disc touches the wall (or the wall's end), and of several
    walls' answers the slowest is taken.  A cheap vectorised test finds the (agent, segment) pairs that can touch at all; only
    those run the exact scalar geometry."""
    pos, action = np.asarray(pos, np.float64), np.asarray(action, np.float64)
    segments = np.asarray(segments, np.float64).reshape(-1, 2, 2)
    out = action.copy()
    B = pos.shape[0]
    if B == 0 or len(segments) == 0:
        return out
    radius = np.broadcast_to(np.asarray(radius, np.float64), (B,))
    mid = pos + 0.5 * time_step * action
    reach = 0.5 * time_step * np.sqrt(action[:, 0] ** 2 + action[:, 1] ** 2) + radius + 1e-6
    d = _point_segment_dist(segments[None, :, 0, 0], segments[None, :, 0, 1], segments[None, :, 1, 0], segments[None, :, 1, 1],
                            mid[:, None, 0], mid[:, None, 1])
    near = d < reach[:, None]
    for b in np.nonzero(near.any(axis=1))[0]:
        out[b] = _constrain_one(pos[b], action[b], float(radius[b]), time_step, segments[near[b]])
    return out


# ------------------------------------------------------------------------------------------------ step outcomes
def constrain_unicycle_actions(pos, theta, v, r_step, radius, time_step: float, segments) -> np.ndarray:
    """``constrain_agent_action_exact`` for B non-holonomic agents (the MPC's robot: ``ActionRot(v, r)``, the step goes along heading
    theta + r with signed speed v, ``agent_plus.py:175-185``): the speeds actually driven [B] - the rotation is kept as commanded
    (``crowd_sim_plus.py:976-987``).  Pinned by ``tests/golden/env_rotconstrain_*.npz``."""
    pos = np.asarray(pos, np.float64)
    B = pos.shape[0]
    theta, v, r_step = (np.broadcast_to(np.asarray(a, np.float64), (B,)) for a in (theta, v, r_step))
    segments = np.asarray(segments, np.float64).reshape(-1, 2, 2)
    out = np.array(v, np.float64)
    if B == 0 or len(segments) == 0:
        return out
    radius = np.broadcast_to(np.asarray(radius, np.float64), (B,))
    vel = np.stack([np.cos(theta + r_step) * v, np.sin(theta + r_step) * v], axis=1)
    mid = pos + 0.5 * time_step * vel
    reach = 0.5 * time_step * np.abs(v) + radius + 1e-6
    d = _point_segment_dist(segments[None, :, 0, 0], segments[None, :, 0, 1], segments[None, :, 1, 0], segments[None, :, 1, 1],
                            mid[:, None, 0], mid[:, None, 1])
    near = d < reach[:, None]
    for b in np.nonzero(near.any(axis=1))[0]:
        out[b] = _constrain_one(pos[b], None, float(radius[b]), time_step, segments[near[b]],
                                rot=(float(theta[b]), float(v[b]), float(r_step[b])))
    return out


INFO_KEYS = ("ReachGoal", "Timeout", "Collision", "WallCollision", "Frozen", "Danger", "Progress", "AngularSmoothness",
             "LinearSmoothness")


def shipped_rewards() -> Dict[str, float]:
    """``[reward]`` of env.config:66-71 as ``configure()`` leaves it for a non-RL policy (``crowd_sim_plus.py:92-128``): the listed
    terms, the discomfort switch, and -1 for the terms a test run must be able to detect."""
    return {"success_reward": 1.0, "collision_penalty": -0.25, "freezing_penalty": -0.125, "discomfort_dist": 0.2,
            "discomfort_penalty_factor": 0.5, "discomfort": True, "timeout": -1.0, "wall_collision_penalty": -1.0}


def step_outcomes(robot_pos, robot_action, robot_goal, robot_radius, human_pos, human_action, human_radius, global_time,
                  time_limit: float, time_step: float, rewards: Dict[str, float], stat_collision=None, prev_dist_to_goal=None,
                  prev_angular=None, prev_linear=None, detailed: bool = False, robot_rot=None) -> Dict[str, np.ndarray]:
    """The outcome block of ``CrowdSimPlus.step`` (``crowd_sim_plus.py:1067-1166``) for E episodes at once.  Holonomic robot:
    ``robot_action`` [E, 2] is its velocity.  The MPC's unicycle (``sicnav_acados.py:143``): ``robot_rot`` = (theta [E], v [E], r [E]) -
    ``ActionRot(v, r)`` on heading theta, the step goes along theta + r (``agent_plus.py:175-185``; ``robot_action`` is ignored); "frozen"
    tests |v|, the angular term is |r| * time_step, the linear term compares the SIGNED speeds (``:1085-1088, 1144-1164``).

    robot_pos, robot_action, robot_goal [E, 2] (the action already wall-constrained), robot_radius [E] or scalar; human_pos,
    human_action [E, N, 2], human_radius [E, N]; global_time [E] or scalar (BEFORE the step); stat_collision [E] (did the walls
    change the robot's action).  ``prev_*``: the state the smoothness / progress terms carry between steps (NaN = first step).
    Returns per episode: ``collision`` (a human's end-of-step position within the two radii of the robot's; as in the reference
    the scan stops at the first colliding human, so ``dmin`` is the smallest distance among the humans BEFORE it), ``dmin``,
    ``frozen`` (the step is shorter than 1 cm), ``reached_goal``, ``timeout``, ``done``, ``reward``, the value of every info term
    (``info[<key>]``, 0 where it did not fire) and the carried state (``next_prev_*``)."""
    if robot_rot is not None:
        th, rv, rw = (np.asarray(a, np.float64) for a in robot_rot)
        head = th + rw
        robot_action = np.stack([np.cos(head) * rv, np.sin(head) * rv], axis=1)
    rp, ra, rg = (np.asarray(a, np.float64) for a in (robot_pos, robot_action, robot_goal))
    hp, ha, hr = (np.asarray(a, np.float64) for a in (human_pos, human_action, human_radius))
    E = rp.shape[0]
    rr = np.broadcast_to(np.asarray(robot_radius, np.float64), (E,))
    gt = np.broadcast_to(np.asarray(global_time, np.float64), (E,))
    has = lambda k: detailed or k in rewards
    end = rp + ra * time_step
    hend = hp + ha * time_step
    dist = _norm(end[:, None, :] - hend)                                          # [E, N]
    hit = dist < (rr[:, None] + hr)
    collision = hit.any(axis=1)
    first = np.where(collision, hit.argmax(axis=1), hit.shape[1])
    before = np.arange(hit.shape[1])[None, :] < first[:, None]
    dmin = np.where(before, dist, np.inf).min(axis=1, initial=np.inf)
    frozen = (np.sqrt(ra[:, 0] ** 2 + ra[:, 1] ** 2) * time_step < 0.01) if robot_rot is None else (np.abs(rv * time_step) < 0.01)
    reached = _norm(end - rg) < rr
    curr_dist = _norm(rg - end)
    info = {k: np.zeros(E) for k in INFO_KEYS}
    reward = np.zeros(E)
    done = np.zeros(E, dtype=bool)
    goal_fires = reached if has("success_reward") else np.zeros(E, dtype=bool)
    if has("success_reward"):
        info["ReachGoal"] = np.where(goal_fires, rewards["success_reward"], 0.0)
        reward += info["ReachGoal"]
        done |= goal_fires
    timeout = ~goal_fires & (gt >= time_limit)
    if has("timeout"):
        info["Timeout"] = np.where(timeout, rewards["timeout"], 0.0)
        reward += info["Timeout"]
    done |= timeout
    if has("collision_penalty"):
        info["Collision"] = np.where(collision, rewards["collision_penalty"], 0.0)
        reward += info["Collision"]
    sc = np.zeros(E, dtype=bool) if stat_collision is None else np.asarray(stat_collision, bool)
    if has("wall_collision_penalty"):
        info["WallCollision"] = np.where(sc, rewards["wall_collision_penalty"], 0.0)
        reward += info["WallCollision"]
    danger = (dmin < rewards["discomfort_dist"]) if (detailed or rewards.get("discomfort") is True) else np.zeros(E, dtype=bool)
    if danger.any():
        info["Danger"] = np.where(danger, (np.where(danger, dmin, 0.0) - rewards["discomfort_dist"]) * rewards["discomfort_penalty_factor"]
                                  * time_step, 0.0)
        reward += info["Danger"]
    nan = np.full(E, np.nan)
    next_dist = nan.copy() if prev_dist_to_goal is None else np.asarray(prev_dist_to_goal, np.float64).copy()
    if has("progress_factor"):
        info["Progress"] = (next_dist - curr_dist) * rewards["progress_factor"]
        reward += info["Progress"]
        next_dist = curr_dist.copy()
    if has("freezing_penalty"):
        info["Frozen"] = np.where(frozen, rewards["freezing_penalty"], 0.0)
        reward += info["Frozen"]
    next_ang = nan.copy() if prev_angular is None else np.asarray(prev_angular, np.float64).copy()
    if has("angular_smoothness_factor"):
        cur = np.arctan2(ra[:, 1], ra[:, 0]) if robot_rot is None else rw
        firststep = np.isnan(next_ang)
        diff = np.abs(cur - np.where(firststep, cur, next_ang)) if robot_rot is None else cur * time_step      # (a point turn: the angle itself)
        info["AngularSmoothness"] = np.where(firststep, 0.0, np.abs(diff) * rewards["angular_smoothness_factor"])
        reward += info["AngularSmoothness"]
        next_ang = cur
    next_lin = nan.copy() if prev_linear is None else np.asarray(prev_linear, np.float64).copy()
    if has("linear_smoothness_factor"):
        cur = np.sqrt(ra[:, 0] ** 2 + ra[:, 1] ** 2) if robot_rot is None else rv
        firststep = np.isnan(next_lin)
        info["LinearSmoothness"] = np.where(firststep, 0.0, np.abs(np.where(firststep, cur, next_lin) - cur)
                                            * rewards["linear_smoothness_factor"])
        reward += info["LinearSmoothness"]
        next_lin = cur
    return dict(collision=collision, dmin=dmin, frozen=frozen, reached_goal=reached, timeout=timeout, done=done, reward=reward,
                curr_dist_to_goal=curr_dist, info=info, next_prev_dist=next_dist, next_prev_angular=next_ang,
                next_prev_linear=next_lin)


# ------------------------------------------------------------------------------------------------ ORCA with walls (orca_plus)
RVO_EPSILON = 0.00001


def _det(a, b):
    return a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]


def _dot(a, b):
    return a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]


def orca_plus_parameters(pos, goal, radius, v_pref, safety_space: float = 0.05, time_step: float = 0.25) -> Dict[str, np.ndarray]:
    """What ``ORCAPlus.predict`` hands to rvo2 for an agent's step (``policy/orca_plus.py:43-84``; parameters of ``orca.py:56-67``,
    radius and safety space from ``[humans]`` through ``ORCAPlus.configure``): the inflated radius, the ego's speed limit (its
    v_pref), the preferred velocity - goal minus position, capped at v_pref - 1e-3 (NOT at 1 as plain ORCA does) - and the scalars.
    Arrays of any leading shape.  Pinned by ``tests/golden/env_orca_plus_calls_*.npz``."""
    to_goal = np.asarray(goal, np.float64) - np.asarray(pos, np.float64)
    speed = _norm(to_goal)
    cap = np.asarray(v_pref, np.float64) - 1e-3
    with np.errstate(invalid="ignore", divide="ignore"):
        pref = np.where((speed > cap)[..., None], to_goal / speed[..., None] * cap[..., None], to_goal)
    return dict(radius=np.asarray(radius, np.float64) + 0.01 + safety_space, max_speed=np.asarray(v_pref, np.float64), pref=pref,
                neighbor_dist=10.0, max_neighbors=10, time_horizon=2.0, time_horizon_obst=0.5, time_step=time_step,
                default_radius=0.20, default_max_speed=1.0)


def obstacle_orca_lines(pos, vel, radius, max_speed, segments, time_horizon_obst: float = 0.5):
    """RVO2's obstacle half-planes (``Agent::computeNewVelocity``, first loop; RVO2 Library 2.0.2) for B agents against L wall
    segments, each wall a two-vertex obstacle as ``orca_plus.py:52-55`` adds them: pos, vel [B, 2], radius, max_speed [B] ->
    (point, direction) [B, 2 L, 2] and ``valid`` [B, 2 L], the accepted lines compacted to the left in RVO2's order (nearest
    edge first; an edge whose velocity obstacle earlier lines already cover adds none).  UNPINNED (rvo2 is absent); equals the
    scalar, one-agent-at-a-time restatement kept with the test infrastructure (tests/test_crowd_env.py)."""
    pos, vel = np.asarray(pos, np.float64), np.asarray(vel, np.float64)
    B = pos.shape[0]
    segments = np.asarray(segments, np.float64).reshape(-1, 2, 2)
    L2 = 2 * len(segments)
    P_out, D_out, valid = np.zeros((B, L2, 2)), np.tile(np.array([1.0, 0.0]), (B, L2, 1)), np.zeros((B, L2), dtype=bool)
    if B == 0 or L2 == 0:
        return P_out, D_out, valid
    radius = np.broadcast_to(np.asarray(radius, np.float64), (B,))[:, None]
    max_speed = np.broadcast_to(np.asarray(max_speed, np.float64), (B,))[:, None]
    # directed edges: A -> B of every wall, then B -> A
    p1 = np.concatenate([segments[:, 0], segments[:, 1]])[None]          # [1, 2L, 2]
    p2 = np.concatenate([segments[:, 1], segments[:, 0]])[None]
    ov = p2 - p1
    ov_sq = _dot(ov, ov)
    u1 = ov / np.sqrt(ov_sq)[..., None]
    u2 = -u1
    x = pos[:, None, :]
    v = vel[:, None, :]
    inv_t = 1.0 / time_horizon_obst
    r_sq = radius * radius
    rp1, rp2 = p1 - x, p2 - x                                              # [B, 2L, 2]
    # ---- neighbours: agent on the right of the edge, edge within reach; nearest first
    range_sq = (time_horizon_obst * max_speed + radius) ** 2
    left_of = _det(rp1, ov)
    rr = _dot(-rp1, ov) / ov_sq
    q = np.where((rr < 0.0)[..., None], -rp1, np.where((rr > 1.0)[..., None], -rp2, -rp1 - rr[..., None] * ov))
    dseg = _dot(q, q)
    nb = (left_of < 0.0) & (left_of * left_of / ov_sq < range_sq) & (dseg < range_sq)
    # ---- the half-plane every edge WOULD contribute (all branches, masked)
    d1, d2 = _dot(rp1, rp1), _dot(rp2, rp2)
    s = rr
    ql = -rp1 - s[..., None] * ov
    d_line = _dot(ql, ql)
    perp = lambda a: np.stack([-a[..., 1], a[..., 0]], axis=-1)
    with np.errstate(invalid="ignore", divide="ignore"):
        unit = lambda a: a / np.sqrt(_dot(a, a))[..., None]
        c1 = (s < 0.0) & (d1 <= r_sq)
        c2 = ~c1 & (s > 1.0) & (d2 <= r_sq)
        c3 = ~c1 & ~c2 & (s >= 0.0) & (s < 1.0) & (d_line <= r_sq)
        coll = c1 | c2 | c3
        coll_dir = np.where(c1[..., None], unit(perp(rp1)), np.where(c2[..., None], unit(perp(rp2)), -u1))
        coll_emit = c1 | c3 | (c2 & (_det(rp2, u2) >= 0.0))
        obl_l = ~coll & (s < 0.0) & (d_line <= r_sq)
        obl_r = ~coll & (s > 1.0) & (d_line <= r_sq)
        same = obl_l | obl_r
        leg = lambda rp, dd, sign: np.stack([rp[..., 0] * np.sqrt(dd - r_sq) - sign * rp[..., 1] * radius,
                                             sign * rp[..., 0] * radius + rp[..., 1] * np.sqrt(dd - r_sq)], axis=-1) / dd[..., None]
        left1, right1 = leg(rp1, d1, 1.0), leg(rp1, d1, -1.0)
        left2, right2 = leg(rp2, d2, 1.0), leg(rp2, d2, -1.0)
        left = np.where(obl_r[..., None], left2, left1)
        right = np.where(obl_l[..., None], right1, right2)
        o1rel = np.where(obl_r[..., None], rp2, rp1)                     # "obstacle1" - position, after the oblique cases
        o2rel = np.where(obl_l[..., None], rp1, rp2)
        u_right_vertex = np.where(obl_l[..., None], u1, u2)
        u_o1 = np.where(obl_r[..., None], u2, u1)
        u_prev = np.where(obl_r[..., None], u1, u2)
        left_foreign = _det(left, -u_prev) >= 0.0
        left = np.where(left_foreign[..., None], -u_prev, left)
        right_foreign = _det(right, u_right_vertex) <= 0.0
        right = np.where(right_foreign[..., None], u_right_vertex, right)
        lc, rc = inv_t * o1rel, inv_t * o2rel
        cv = rc - lc
        t = np.where(same, 0.5, _dot(v - lc, cv) / _dot(cv, cv))
        t_left, t_right = _dot(v - lc, left), _dot(v - rc, right)
        g1 = ((t < 0.0) & (t_left < 0.0)) | (same & (t_left < 0.0) & (t_right < 0.0))
        g2 = ~g1 & (t > 1.0) & (t_right < 0.0)
        uw1, uw2 = unit(v - lc), unit(v - rc)
        inf = np.inf
        c = v - (lc + t[..., None] * cv)
        dsq_cut = np.where((t < 0.0) | (t > 1.0) | same, inf, _dot(c, c))
        c = v - (lc + t_left[..., None] * left)
        dsq_left = np.where(t_left < 0.0, inf, _dot(c, c))
        c = v - (rc + t_right[..., None] * right)
        dsq_right = np.where(t_right < 0.0, inf, _dot(c, c))
        on_cut = ~g1 & ~g2 & (dsq_cut <= dsq_left) & (dsq_cut <= dsq_right)
        on_left = ~g1 & ~g2 & ~on_cut & (dsq_left <= dsq_right)
        on_right = ~g1 & ~g2 & ~on_cut & ~on_left
        gen_dir = np.where(g1[..., None], np.stack([uw1[..., 1], -uw1[..., 0]], -1),
                  np.where(g2[..., None], np.stack([uw2[..., 1], -uw2[..., 0]], -1),
                  np.where(on_cut[..., None], -u_o1, np.where(on_left[..., None], left, -right))))
        off = radius[..., None] * inv_t
        gen_pt = np.where(g1[..., None], lc + off * uw1,
                 np.where(g2[..., None], rc + off * uw2,
                 np.where((on_cut | on_left)[..., None], lc + off * perp(gen_dir), rc + off * perp(gen_dir))))
        gen_emit = g1 | g2 | on_cut | (on_left & ~left_foreign) | (on_right & ~right_foreign)
    cand_pt = np.where(coll[..., None], 0.0, gen_pt)
    cand_dir = np.where(coll[..., None], coll_dir, gen_dir)
    cand_emit = np.where(coll, coll_emit, gen_emit)
    # ---- RVO2's order: nearest edge first; an edge already covered by the accepted lines adds nothing
    key = np.where(nb, dseg, np.inf)
    order = np.argsort(key, axis=1, kind="stable")
    count = np.zeros(B, dtype=np.int64)
    rows = np.arange(B)
    slot = np.arange(L2)[None, :]
    for k in range(L2):
        e = order[:, k]
        live = nb[rows, e]
        if not live.any():
            break
        a1, a2 = inv_t * rp1[rows, e], inv_t * rp2[rows, e]                # [B, 2]
        ok1 = _det(a1[:, None, :] - P_out, D_out) - inv_t * radius >= -RVO_EPSILON
        ok2 = _det(a2[:, None, :] - P_out, D_out) - inv_t * radius >= -RVO_EPSILON
        covered = (ok1 & ok2 & (slot < count[:, None])).any(axis=1)
        take = live & ~covered & cand_emit[rows, e]
        dst = np.where(take, count, 0)
        P_out[rows[take], dst[take]] = cand_pt[rows[take], e[take]]
        D_out[rows[take], dst[take]] = cand_dir[rows[take], e[take]]
        valid[rows[take], dst[take]] = True
        count = count + take
    return P_out, D_out, valid


def orca_plus_velocities(pos, vel, radius, pref, max_speed, segments, time_horizon: float = 2.0, time_horizon_obst: float = 0.5,
                         time_step: float = 0.25, neighbor_dist: float = 10.0, max_neighbors: int = 10) -> np.ndarray:
    """New ORCA velocity of EVERY agent of every episode among the other agents of its episode AND the walls, each as the ego of
    its own program (``ORCAPlus.predict``): pos, vel, pref [E, n, 2]; radius, max_speed [E, n] (radius already inflated);
    segments [L, 2, 2] shared by all episodes.  Obstacle half-planes come first and stay hard constraints when the program is
    infeasible (``linearProgram3`` with ``numObstLines``).  -> [E, n, 2]."""
    from . import episodes as EP
    E, n, _ = pos.shape
    B = E * n
    ego = lambda a: a.reshape((B,) + a.shape[2:])
    Po, Do, vo = obstacle_orca_lines(ego(pos), ego(vel), ego(radius), ego(max_speed), segments, time_horizon_obst)
    if n > 1:
        others = np.array([[j for j in range(n) if j != i] for i in range(n)], dtype=np.int64).reshape(n, n - 1)
        oth = lambda a: a[:, others].reshape((B, n - 1) + a.shape[2:])
        Pa, Da, dsq = EP.orca_lines(ego(pos), ego(vel), ego(radius), oth(pos), oth(vel), oth(radius), time_horizon, time_step,
                                    with_dist=True)
        va = (dsq < neighbor_dist * neighbor_dist) & (np.arange(n - 1)[None, :] < max_neighbors)
    else:
        Pa, Da, va = np.zeros((B, 0, 2)), np.zeros((B, 0, 2)), np.zeros((B, 0), dtype=bool)
    P, D, active = np.concatenate([Po, Pa], 1), np.concatenate([Do, Da], 1), np.concatenate([vo, va], 1)
    is_obst = np.concatenate([vo, np.zeros_like(va)], 1)
    off = ~active            # slots that hold no line: a half-plane every velocity satisfies (the programs skip inactive lines)
    D = np.where(off[..., None], np.array([1.0, 0.0]), D)
    P = np.where(off[..., None], np.array([0.0, -1e9]), P)
    allrows = np.ones(B, dtype=bool)
    failed, fail_idx, result = EP._lp2(P, D, active, ego(max_speed), ego(pref), False, allrows)
    if failed.any():
        result = EP._lp3(P, D, fail_idx, ego(max_speed), result, failed, is_obst=is_obst)
    return result.reshape(E, n, 2)


# ------------------------------------------------------------------------------------------------ social-force humans (sfm)
@dataclass(frozen=True)
class SFMParams:
    """``[humans]`` of env.config as ``SFM.configure(config, 'humans')`` reads it (``policy/social_force.py:21-36``)."""
    radius: float = 0.20
    A: float = 3.0
    B: float = 0.18
    KI: float = 1.0
    A_static: float = 2.0
    B_static: float = 0.025
    A_bottleneck: float = 6.0
    B_bottleneck: float = 0.12


def sfm_velocities(pos, vel, goal, radius, v_pref, others_pos, others_radius, segments, time_step: float = 0.25,
                   params: SFMParams = SFMParams(), bottleneck: bool = False) -> np.ndarray:
    """``SFM.predict`` (``policy/social_force.py:38-95``) for B agents at once: a pull towards the goal at the preferred speed
    (relaxation rate KI), an exponential push away from every other agent (``others_pos`` [B, m, 2], ``others_radius`` [B, m], in
    the order the reference sums them: the other humans, then the robot) and from the closest point of every wall (in the order
    of the rule's segments; from the third segment on with the bottleneck constants when ``bottleneck``), integrated over one
    step and clipped to the preferred speed.  pos, vel, goal [B, 2]; radius, v_pref [B] -> [B, 2].  The sums run in the
    reference's order, term by term.  Pinned by ``tests/golden/env_sfm_calls_*.npz`` and, through whole episodes, by
    ``tests/golden/env_rollout_sfm_*.npz``."""
    pos, vel, goal = (np.asarray(a, np.float64) for a in (pos, vel, goal))
    radius, v_pref = np.asarray(radius, np.float64), np.asarray(v_pref, np.float64)
    others_pos, others_radius = np.asarray(others_pos, np.float64), np.asarray(others_radius, np.float64)
    segments = np.asarray(segments, np.float64).reshape(-1, 2, 2)
    dx, dy = goal[:, 0] - pos[:, 0], goal[:, 1] - pos[:, 1]
    dist = np.sqrt(dx ** 2 + dy ** 2)
    dist = np.where(dist < 1e-6, 1.0, dist)
    dvx = params.KI * ((dx / dist) * v_pref - vel[:, 0])
    dvy = params.KI * ((dy / dist) * v_pref - vel[:, 1])
    ivx, ivy = np.zeros(len(pos)), np.zeros(len(pos))
    for j in range(others_pos.shape[1]):
        adj = np.abs(params.radius - others_radius[:, j]) + 0.01
        ox, oy = pos[:, 0] - others_pos[:, j, 0], pos[:, 1] - others_pos[:, j, 1]
        d = np.sqrt(ox ** 2 + oy ** 2)
        ivx = ivx + params.A * np.exp((radius + others_radius[:, j] + adj - d) / params.B) * (ox / d)
        ivy = ivy + params.A * np.exp((radius + others_radius[:, j] + adj - d) / params.B) * (oy / d)
    for idx, sg in enumerate(segments):
        a_s, b_s = (params.A_bottleneck, params.B_bottleneck) if (bottleneck and idx >= 2) else (params.A_static, params.B_static)
        px, py = sg[1, 0] - sg[0, 0], sg[1, 1] - sg[0, 1]
        u = ((pos[:, 0] - sg[0, 0]) * px + (pos[:, 1] - sg[0, 1]) * py) / (px * px + py * py)
        u = np.where(u > 1, 1.0, np.where(u < 0, 0.0, u))
        ox, oy = pos[:, 0] - (sg[0, 0] + u * px), pos[:, 1] - (sg[0, 1] + u * py)
        d = np.sqrt(ox ** 2 + oy ** 2)
        ivx = ivx + a_s * np.exp((radius + 0.01 - d) / b_s) * (ox / d)
        ivy = ivy + a_s * np.exp((radius + 0.01 - d) / b_s) * (oy / d)
    nvx = vel[:, 0] + (dvx + ivx) * time_step
    nvy = vel[:, 1] + (dvy + ivy) * time_step
    nrm = _norm(np.stack([nvx, nvy], axis=-1))
    with np.errstate(invalid="ignore", divide="ignore"):
        clip = nrm > v_pref
        return np.stack([np.where(clip, nvx / nrm * v_pref, nvx), np.where(clip, nvy / nrm * v_pref, nvy)], axis=-1)


# ------------------------------------------------------------------------------------------------ hallway episodes
@dataclass
class HallwayConfig:
    """The fields of ``sicnav_diffusion/configs/env.config`` a hallway episode reads (shipped values)."""
    rule: str = "hallway"                   # [sim] test_sim
    geometry: Geometry = Geometry()         # [sim] circle_radius, rect_width, rect_height; [robot] radius
    time_step: float = 0.25                 # [env] time_step
    time_limit: float = 30.0                # [env] time_limit
    starts_moving: int = 10                 # [sim] starts_moving: steps the humans walk before the robot's clock starts
    human_radius: float = 0.20              # [humans] radius
    human_v_pref: float = 1.5               # [humans] v_pref (drawn from U(0.5, 1.5) with randomize_attributes)
    safety_space: float = 0.05              # [humans] safety_space
    robot_v_pref: float = 1.0               # [robot] v_pref
    randomize_attributes: bool = True       # [env] randomize_attributes
    discomfort_dist: float = 0.2            # [reward] discomfort_dist
    human_policy: str = "orca_plus"         # [humans] policy: "orca_plus" (shipped; rvo2's solve is unpinned) or "sfm" (pinned end to end)
    sfm: SFMParams = SFMParams()            # [humans] A, B, KI, A_static, ... (the sfm policy's constants)


def reference_case_seed(phase: str, case: int, val_size: int = 100, test_size: int = 500) -> int:
    """The seed ``reset`` gives the generator of test case ``case`` of a phase (``crowd_sim_plus.py:656-663``: counter_offset =
    {train: val_size + test_size, val: 0, test: val_size} + the case counter; sizes from env.config ``[env] val_size, test_size``)."""
    return {"train": val_size + test_size, "val": 0, "test": val_size}[phase] + int(case)


def hallway_starts(E: int, N: int, seed: int, cfg: HallwayConfig, reference_cases: Optional[str] = None) -> Dict[str, np.ndarray]:
    """Start record of E hallway episodes (``reset``, ``crowd_sim_plus.py:660-672``: robot at (0, -R) heading for (0, R), then the
    walls of the rule, then N humans by ``generate_hallway_human``), every episode on its own generator: by default
    ``episodes.episode_rng(seed, e)``; with ``reference_cases`` = "test" / "val" / "train" episode e IS the reference's case
    ``seed + e`` of that phase - ``np.random.default_rng(reference_case_seed(phase, seed + e))``, the crowd ``reset(phase,
    test_case=seed + e)`` places.  Index 0 of the agent axis is the robot."""
    from .episodes import episode_rng
    if reference_cases is not None:
        episode_rng = lambda sd, e: np.random.default_rng(reference_case_seed(reference_cases, sd + e))      # noqa: E731
    R = cfg.geometry.circle_radius
    pos, goal = np.zeros((E, N + 1, 2)), np.zeros((E, N + 1, 2))
    rad = np.full((E, N + 1), cfg.human_radius)
    vp = np.full((E, N + 1), cfg.human_v_pref)
    pos[:, 0], goal[:, 0] = (0.0, -R), (0.0, R)
    rad[:, 0], vp[:, 0] = cfg.geometry.robot_radius, cfg.robot_v_pref
    for e in range(E):
        h = place_hallway_humans(N, episode_rng(seed, e), cfg.rule, cfg.geometry, cfg.human_radius, cfg.human_v_pref,
                                 cfg.randomize_attributes, cfg.discomfort_dist)
        pos[e, 1:], goal[e, 1:], vp[e, 1:] = h["pos"], h["final_goal"], h["v_pref"]
    return dict(pos=pos, goal=goal, radius=rad, v_pref=vp)


def simulate_hallway(E: int, N: int, steps: int, seed: int, cfg: Optional[HallwayConfig] = None, robot="orca",
                     starts: Optional[Dict[str, np.ndarray]] = None) -> Dict[str, np.ndarray]:
    """E independent episodes of a hallway rule with N ORCA humans that see the walls (``orca_plus``), ``steps`` simulator steps
    after the ``starts_moving`` steps the reference lets the humans walk while the robot stands (``crowd_sim_plus.py:707-721``).

    Per step, as ``CrowdSimPlus.step``: every human's goal is its door sub-goal where one applies, its ORCA velocity among the
    others and the walls (``cfg.human_policy`` = "orca_plus") or its social-force velocity (= "sfm": every term in the reference's
    own arithmetic - whole episodes reproduce the reference's lines, ``tests/golden/env_rollout_sfm_*.npz``) is cut short at the
    walls (``constrain_agent_action_exact``), then all agents move.  The reference's robot is driven by the MPC (out of scope);
    here ``robot`` = "orca" (an ORCA agent like the orca_plus humans, v_pref 1), "goal" (straight at its goal at v_pref), "still",
    an array [E, steps, 2] of commanded velocities (wall-constrained like everybody's), or - the MPC's own kinematics
    (``sicnav_acados.py:143``) - a dict ``{"v": [E, steps], "r": [E, steps]}`` of ``ActionRot`` commands for a unicycle that starts
    heading pi / 2 (``crowd_sim_plus.py:661``): heading and position by ``agent_plus.py:175-214``, the wall constraint and the
    outcome block in their ``ActionRot`` branches; ``robot_theta`` [E, steps + 1] is returned with it.  ``starts``: a start record in the
    place of ``hallway_starts`` (pos, goal [E, N + 1, 2], radius, v_pref [E, N + 1]).  Returns what ``simulate_circle_crossing``
    returns (positions from the moment the robot's clock starts, frame 0 = global time 0) plus per step the outcome block
    (``step_outcomes`` with the shipped rewards): ``collision``, ``dmin``, ``reached_goal``, ``timeout``, ``done``, ``reward`` [E, steps]
    and the step at which each episode ended (``end_step``, -1 = still running), and ``human_times`` [E, N]: the reference's record of
    when a human first stood within its radius of its CURRENT goal - the door's centre while a sub-goal applies - on the
    environment's clock, which starts at -starts_moving * time_step; 0 = not yet (``crowd_sim_plus.py:1203-1206`` with its quirk: a time
    of exactly 0.0 reads as "not yet" and is overwritten by the next arrival test that fires)."""
    cfg = cfg or HallwayConfig()
    segs, doors = static_obstacles(cfg.rule, cfg.geometry)
    st = starts if starts is not None else hallway_starts(E, N, seed, cfg)
    pos, final_goal, radius, v_pref = np.array(st["pos"], np.float64), st["goal"], st["radius"], st["v_pref"]
    if cfg.human_policy not in ("orca_plus", "sfm"):
        raise ValueError("human_policy must be 'orca_plus' or 'sfm'")
    robot_mode = robot if isinstance(robot, str) else "unicycle" if isinstance(robot, dict) else "given"
    if robot_mode not in ("orca", "goal", "still", "given", "unicycle"):
        raise ValueError("robot must be 'orca', 'goal', 'still', an array of commanded velocities or a dict of ActionRot commands")
    theta = np.full(E, np.pi / 2)
    thetas = np.zeros((E, steps + 1))
    vel = np.zeros_like(pos)
    dt = cfg.time_step
    rewards = shipped_rewards()
    total = cfg.starts_moving + steps
    traj, vels = np.zeros((E, steps + 1, N + 1, 2)), np.zeros((E, steps + 1, N + 1, 2))
    out = {k: np.zeros((E, steps), dtype=t) for k, t in (("collision", bool), ("dmin", float), ("reached_goal", bool), ("timeout", bool),
                                                          ("done", bool), ("reward", float), ("wall_collision", bool))}
    end_step = np.full(E, -1)
    human_times = np.zeros((E, N))
    global_time = -cfg.starts_moving * dt                      # crowd_sim_plus.py:712
    goal = final_goal.copy()
    goal[:, 1:] = door_subgoal(pos[:, 1:], final_goal[:, 1:], cfg.rule, doors, len(segs))
    for s in range(total):
        live = s >= cfg.starts_moving
        new_vel = np.zeros_like(vel)
        if cfg.human_policy == "orca_plus" or (live and robot_mode == "orca"):
            par = orca_plus_parameters(pos, goal, radius, v_pref, cfg.safety_space, dt)
            orca = orca_plus_velocities(pos, vel, par["radius"], par["pref"], par["max_speed"], segs, par["time_horizon"],
                                        par["time_horizon_obst"], dt, par["neighbor_dist"], par["max_neighbors"])
            new_vel[:, 0] = orca[:, 0]
            if cfg.human_policy == "orca_plus":
                new_vel[:, 1:] = orca[:, 1:]
        if cfg.human_policy == "sfm":
            # every human's observation: the other humans in list order, then the robot (crowd_sim_plus.py:1044-1052)
            idx = np.array([[j for j in range(1, N + 1) if j != i] + [0] for i in range(1, N + 1)], dtype=np.int64).reshape(N, N)
            flat = lambda a: a.reshape((E * N,) + a.shape[2:])
            new_vel[:, 1:] = sfm_velocities(flat(pos[:, 1:]), flat(vel[:, 1:]), flat(goal[:, 1:]), flat(radius[:, 1:]), flat(v_pref[:, 1:]),
                                            flat(pos[:, idx]), flat(radius[:, idx]), segs, dt, cfg.sfm,
                                            cfg.rule == "hallway_bottleneck").reshape(E, N, 2)
        if not live or robot_mode == "still":
            new_vel[:, 0] = 0.0                                       # the dummy start: ActionXY(0, 0) for the robot
        elif robot_mode == "given":
            new_vel[:, 0] = np.asarray(robot, np.float64)[:, s - cfg.starts_moving]
        elif robot_mode == "unicycle":
            v_cmd = np.asarray(robot["v"], np.float64)[:, s - cfg.starts_moving]
            r_cmd = np.asarray(robot["r"], np.float64)[:, s - cfg.starts_moving]
            v_drv = constrain_unicycle_actions(pos[:, 0], theta, v_cmd, r_cmd, radius[:, 0], dt, segs)
            head = theta + r_cmd
            new_vel[:, 0] = np.stack([np.cos(head) * v_drv, np.sin(head) * v_drv], axis=1)
        elif robot_mode == "goal":
            d = final_goal[:, 0] - pos[:, 0]
            new_vel[:, 0] = d / np.maximum(_norm(d), 1e-9)[:, None] * v_pref[:, 0:1]
        wanted = new_vel.copy()
        uni = live and robot_mode == "unicycle"
        new_vel = constrain_actions(pos.reshape(-1, 2), new_vel.reshape(-1, 2), radius.reshape(-1), dt, segs).reshape(E, N + 1, 2)
        if uni:
            new_vel[:, 0] = wanted[:, 0]                               # (cut short above, in its own ActionRot branch)
        wall_hit = (v_cmd != v_drv) if uni else (wanted[:, 0] != new_vel[:, 0])[:, 0]      # crowd_sim_plus.py:1059-1063
        if live:
            k = s - cfg.starts_moving
            if k == 0:
                traj[:, 0], vels[:, 0], thetas[:, 0] = pos, vel, theta
                prev_dist = _norm(final_goal[:, 0] - pos[:, 0])
            o = step_outcomes(pos[:, 0], new_vel[:, 0], final_goal[:, 0], radius[:, 0], pos[:, 1:], new_vel[:, 1:], radius[:, 1:],
                              k * dt, cfg.time_limit, dt, rewards, stat_collision=wall_hit,
                              prev_dist_to_goal=prev_dist, robot_rot=(theta, v_drv, r_cmd) if uni else None)
            for key in ("collision", "dmin", "reached_goal", "timeout", "done", "reward"):
                out[key][:, k] = o[key]
            out["wall_collision"][:, k] = wall_hit
            end_step = np.where((end_step < 0) & o["done"], k, end_step)
        vel = new_vel
        pos = pos + vel * dt
        if uni:                                                    # Agent.step of a unicycle (agent_plus.py:211-214)
            unwrapped = (theta + r_cmd) % (2 * np.pi)
            theta = np.where(unwrapped > np.pi, unwrapped - 2 * np.pi, unwrapped)
            vel[:, 0] = np.stack([v_drv * np.cos(theta), v_drv * np.sin(theta)], axis=1)
        goal = final_goal.copy()                                   # Human.step: set_g_xy at the new position (human_plus.py:19-79)
        goal[:, 1:] = door_subgoal(pos[:, 1:], final_goal[:, 1:], cfg.rule, doors, len(segs))
        global_time += dt
        arrived = _norm((pos[:, 1:] - goal[:, 1:]).reshape(-1, 2)).reshape(E, N) < radius[:, 1:]      # agent_plus.py:217
        human_times = np.where((human_times == 0) & arrived, global_time, human_times)
        if live:
            traj[:, s - cfg.starts_moving + 1], vels[:, s - cfg.starts_moving + 1] = pos, vel
            thetas[:, s - cfg.starts_moving + 1] = theta
    out["human_times"] = human_times
    if robot_mode == "unicycle":
        out["robot_theta"] = thetas
    return dict(human_xy=traj[:, :, 1:], robot_xy=traj[:, :, 0], human_vel=vels[:, :, 1:], stamps=np.arange(steps + 1) * dt,
                goal=final_goal, radius=radius, v_pref=v_pref, segments=segs, end_step=end_step, **out)
