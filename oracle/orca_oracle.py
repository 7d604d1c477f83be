"""CPU ORACLE for the batched episode generator  --  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference's humans are driven by Python-RVO2 (``crowd_sim_plus/envs/policy/orca.py:82-133``
builds an ``rvo2.PyRVOSimulator`` per step and takes agent 0's new velocity); rvo2 is an un-vendored C++ dependency
(requirements: ``Python-RVO2``, RVO2 Library v2.0.2) that is absent from /root/reference and from this image, so
nothing here can be checked against the reference's own output.  This file restates the PUBLISHED algorithm
(van den Berg, Guy, Lin, Manocha: "Reciprocal n-body collision avoidance", ISRR 2011; RVO2 ``Agent::computeNewVelocity``
agent-agent part with ``linearProgram1/2/3``) as plain scalar Python, one agent at a time, in float64 (RVO2 itself is
float32).  The product's vectorised implementation (``safe-interactive-crowdnav_amd/episodes.py``) is held to THIS
restatement by ``tests/test_episodes.py``, which also checks the restatement against a brute-force search of the
velocity disc and against collision-freeness of the resulting crowds.

Call sites anchored: ORCA parameters and call pattern ``orca.py:56-66, 96-131`` (neighbour distance 10, 10 neighbours,
time horizon 2.0, radius + 0.01, ego max speed = v_pref, others' max speed 1, preferred velocity = goal - position
clipped to unit length, other agents' preferred velocity (0, 0)); circle-crossing placement
``crowd_sim_plus.py:454-481``.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

RVO_EPSILON = 0.00001
Vec = Tuple[float, float]
Line = Tuple[Vec, Vec]          # (point, direction)


def det(a: Vec, b: Vec) -> float:
    return a[0] * b[1] - a[1] * b[0]


def dot(a: Vec, b: Vec) -> float:
    return a[0] * b[0] + a[1] * b[1]


def orca_lines(pos: Vec, vel: Vec, radius: float, others: Sequence[Tuple[Vec, Vec, float]], time_horizon: float,
               time_step: float) -> List[Line]:
    """Half-planes of permitted velocities, one per neighbour, nearest neighbour first (RVO2 keeps its neighbour list
    sorted by distance)."""
    inv_th = 1.0 / time_horizon
    order = sorted(range(len(others)), key=lambda i: (others[i][0][0] - pos[0]) ** 2 + (others[i][0][1] - pos[1]) ** 2)
    lines = []
    for i in order:
        op, ov, orad = others[i]
        rp = (op[0] - pos[0], op[1] - pos[1])
        rv = (vel[0] - ov[0], vel[1] - ov[1])
        dist_sq = dot(rp, rp)
        cr = radius + orad
        cr_sq = cr * cr
        if dist_sq > cr_sq:
            w = (rv[0] - inv_th * rp[0], rv[1] - inv_th * rp[1])
            w_len_sq = dot(w, w)
            dp1 = dot(w, rp)
            if dp1 < 0.0 and dp1 * dp1 > cr_sq * w_len_sq:
                w_len = math.sqrt(w_len_sq)
                uw = (w[0] / w_len, w[1] / w_len)
                direction = (uw[1], -uw[0])
                s = cr * inv_th - w_len
                u = (s * uw[0], s * uw[1])
            else:
                leg = math.sqrt(dist_sq - cr_sq)
                if det(rp, w) > 0.0:
                    direction = ((rp[0] * leg - rp[1] * cr) / dist_sq, (rp[0] * cr + rp[1] * leg) / dist_sq)
                else:
                    direction = (-(rp[0] * leg + rp[1] * cr) / dist_sq, -(-rp[0] * cr + rp[1] * leg) / dist_sq)
                dp2 = dot(rv, direction)
                u = (dp2 * direction[0] - rv[0], dp2 * direction[1] - rv[1])
        else:
            inv_ts = 1.0 / time_step
            w = (rv[0] - inv_ts * rp[0], rv[1] - inv_ts * rp[1])
            w_len = math.sqrt(dot(w, w))
            uw = (w[0] / w_len, w[1] / w_len)
            direction = (uw[1], -uw[0])
            s = cr * inv_ts - w_len
            u = (s * uw[0], s * uw[1])
        lines.append(((vel[0] + 0.5 * u[0], vel[1] + 0.5 * u[1]), direction))
    return lines


def linear_program1(lines: Sequence[Line], line_no: int, radius: float, opt: Vec, direction_opt: bool):
    p, d = lines[line_no]
    dp = dot(p, d)
    disc = dp * dp + radius * radius - dot(p, p)
    if disc < 0.0:
        return None
    sq = math.sqrt(disc)
    t_left, t_right = -dp - sq, -dp + sq
    for i in range(line_no):
        pi, di = lines[i]
        den = det(d, di)
        num = det(di, (p[0] - pi[0], p[1] - pi[1]))
        if abs(den) <= RVO_EPSILON:
            if num < 0.0:
                return None
            continue
        t = num / den
        if den >= 0.0:
            t_right = min(t_right, t)
        else:
            t_left = max(t_left, t)
        if t_left > t_right:
            return None
    if direction_opt:
        t = t_right if dot(opt, d) > 0.0 else t_left
    else:
        t = dot(d, (opt[0] - p[0], opt[1] - p[1]))
        t = t_left if t < t_left else (t_right if t > t_right else t)
    return (p[0] + t * d[0], p[1] + t * d[1])


def linear_program2(lines: Sequence[Line], radius: float, opt: Vec, direction_opt: bool):
    if direction_opt:
        result = (opt[0] * radius, opt[1] * radius)
    elif dot(opt, opt) > radius * radius:
        n = math.sqrt(dot(opt, opt))
        result = (opt[0] / n * radius, opt[1] / n * radius)
    else:
        result = opt
    for i, (p, d) in enumerate(lines):
        if det(d, (p[0] - result[0], p[1] - result[1])) > 0.0:
            r = linear_program1(lines, i, radius, opt, direction_opt)
            if r is None:
                return i, result
            result = r
    return len(lines), result


def linear_program3(lines: Sequence[Line], begin_line: int, radius: float, result: Vec) -> Vec:
    distance = 0.0
    for i in range(begin_line, len(lines)):
        pi, di = lines[i]
        if det(di, (pi[0] - result[0], pi[1] - result[1])) > distance:
            proj: List[Line] = []
            for j in range(i):
                pj, dj = lines[j]
                determinant = det(di, dj)
                if abs(determinant) <= RVO_EPSILON:
                    if dot(di, dj) > 0.0:
                        continue
                    point = (0.5 * (pi[0] + pj[0]), 0.5 * (pi[1] + pj[1]))
                else:
                    t = det(dj, (pi[0] - pj[0], pi[1] - pj[1])) / determinant
                    point = (pi[0] + t * di[0], pi[1] + t * di[1])
                dd = (dj[0] - di[0], dj[1] - di[1])
                n = math.sqrt(dot(dd, dd))
                proj.append((point, (dd[0] / n, dd[1] / n)))
            fail, r = linear_program2(proj, radius, (-di[1], di[0]), True)
            if fail >= len(proj):
                result = r
            distance = det(di, (pi[0] - result[0], pi[1] - result[1]))
    return result


def new_velocity(pos: Vec, vel: Vec, radius: float, pref: Vec, max_speed: float,
                 others: Sequence[Tuple[Vec, Vec, float]], time_horizon: float = 2.0, time_step: float = 0.25) -> Vec:
    """One agent's ORCA velocity given the other agents' (position, velocity, radius)."""
    lines = orca_lines(pos, vel, radius, others, time_horizon, time_step)
    fail, result = linear_program2(lines, max_speed, pref, False)
    if fail < len(lines):
        result = linear_program3(lines, fail, max_speed, result)
    return result
