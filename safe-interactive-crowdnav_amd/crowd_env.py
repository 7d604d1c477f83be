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
  ``step_outcomes``           the outcome block of ``step()``       ``crowd_sim_plus.py:1067-1166`` flags, reward terms, done
  ``orca_plus_parameters``    what ``ORCAPlus.predict`` hands rvo2  ``policy/orca_plus.py:46-84``
  ``obstacle_orca_lines``     RVO2's obstacle half-planes           RVO2 Library 2.0.2 ``Agent::computeNewVelocity``

PARITY.  Everything in the first six rows is plain Python / NumPy in the reference and is pinned by fixtures that
``tests/golden/make_golden_env.py`` generates by executing the reference's own lines against stand-in objects
(``tests/golden/env_*.npz``; ``tests/test_crowd_env.py``).  ``obstacle_orca_lines`` restates the C++ of rvo2, an un-vendored
dependency that is absent from the reference tree and from this image: like the agent-agent half-planes of ``episodes.py`` it
is UNPINNED, checked against a scalar restatement kept with the test infrastructure, by brute force and through properties
(no agent of a generated crowd ever enters a wall).  Holonomic agents only (``ActionXY``): the humans always are, the robot
of the shipped configuration is.  The social-force humans (``policy/social_force.py``) are not built.
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


def _constrain_one(cur, act, r, dt, segs):
    """``constrain_agent_action_exact`` for one holonomic agent against the segments that its step comes within r of."""
    fut = cur + act * dt
    move = fut - cur
    move_mag = float(np.sqrt(move @ move))
    hits = []
    for s in segs:
        pA, pB, cd = _closest_between_segments(s[0], s[1], cur, fut)
        if cd - r < 0.0:
            hits.append((s, cd, pA, pB))
    final = act.copy()
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
        cand = (fin - cur) / dt
        if cand[0] ** 2 + cand[1] ** 2 < final[0] ** 2 + final[1] ** 2:       # the slowest of the candidates wins
            final = cand
    return final


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
INFO_KEYS = ("ReachGoal", "Timeout", "Collision", "WallCollision", "Frozen", "Danger", "Progress", "AngularSmoothness",
             "LinearSmoothness")


def shipped_rewards() -> Dict[str, float]:
    """``[reward]`` of env.config:66-71 as ``configure()`` leaves it for a non-RL policy (``crowd_sim_plus.py:92-128``): the listed
    terms, the discomfort switch, and -1 for the terms a test run must be able to detect."""
    return {"success_reward": 1.0, "collision_penalty": -0.25, "freezing_penalty": -0.125, "discomfort_dist": 0.2,
            "discomfort_penalty_factor": 0.5, "discomfort": True, "timeout": -1.0, "wall_collision_penalty": -1.0}


def step_outcomes(robot_pos, robot_action, robot_goal, robot_radius, human_pos, human_action, human_radius, global_time,
                  time_limit: float, time_step: float, rewards: Dict[str, float], stat_collision=None, prev_dist_to_goal=None,
                  prev_angular=None, prev_linear=None, detailed: bool = False) -> Dict[str, np.ndarray]:
    """The outcome block of ``CrowdSimPlus.step`` (``crowd_sim_plus.py:1067-1166``) for E episodes at once, holonomic robot.

    robot_pos, robot_action, robot_goal [E, 2] (the action already wall-constrained), robot_radius [E] or scalar; human_pos,
    human_action [E, N, 2], human_radius [E, N]; global_time [E] or scalar (BEFORE the step); stat_collision [E] (did the walls
    change the robot's action).  ``prev_*``: the state the smoothness / progress terms carry between steps (NaN = first step).
    Returns per episode: ``collision`` (a human's end-of-step position within the two radii of the robot's; as in the reference
    the scan stops at the first colliding human, so ``dmin`` is the smallest distance among the humans BEFORE it), ``dmin``,
    ``frozen`` (the step is shorter than 1 cm), ``reached_goal``, ``timeout``, ``done``, ``reward``, the value of every info term
    (``info[<key>]``, 0 where it did not fire) and the carried state (``next_prev_*``)."""
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
    frozen = np.sqrt(ra[:, 0] ** 2 + ra[:, 1] ** 2) * time_step < 0.01
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
        cur = np.arctan2(ra[:, 1], ra[:, 0])
        firststep = np.isnan(next_ang)
        info["AngularSmoothness"] = np.where(firststep, 0.0, np.abs(np.abs(cur - np.where(firststep, cur, next_ang)))
                                             * rewards["angular_smoothness_factor"])
        reward += info["AngularSmoothness"]
        next_ang = cur
    next_lin = nan.copy() if prev_linear is None else np.asarray(prev_linear, np.float64).copy()
    if has("linear_smoothness_factor"):
        cur = np.sqrt(ra[:, 0] ** 2 + ra[:, 1] ** 2)
        firststep = np.isnan(next_lin)
        info["LinearSmoothness"] = np.where(firststep, 0.0, np.abs(np.where(firststep, cur, next_lin) - cur)
                                            * rewards["linear_smoothness_factor"])
        reward += info["LinearSmoothness"]
        next_lin = cur
    return dict(collision=collision, dmin=dmin, frozen=frozen, reached_goal=reached, timeout=timeout, done=done, reward=reward,
                curr_dist_to_goal=curr_dist, info=info, next_prev_dist=next_dist, next_prev_angular=next_ang,
                next_prev_linear=next_lin)
