#!/usr/bin/env python3
"""Golden capture for the batched episode generator, generated from the REFERENCE'S OWN LINES.

CrowdSimPlus cannot be imported here (gym and rvo2 are absent), and rvo2 itself - the C++ library that solves the ORCA
programs (crowd_sim_plus/envs/policy/orca.py:82-133 calls it) - stays UNPINNABLE: no output of it exists in this container.
What CAN be pinned is everything the reference itself computes around it, which is plain Python / NumPy:
    crowd_sim_plus/envs/crowd_sim_plus.py:454-481   generate_circle_crossing_human: circle placement with positional noise,
                                                    goal at the antipode, rejection against earlier agents' starts and goals
    crowd_sim_plus/envs/crowd_sim_plus.py:484-520   generate_square_crossing_human: start on a random side of the y axis, goal on the
                                                    other side, starts rejected against earlier starts, goals against earlier goals
    crowd_sim_plus/envs/policy/orca.py:56-67        the ORCA parameters (neighbour distance / count, time horizons, radius, speed)
    crowd_sim_plus/envs/policy/orca.py:93-129       what is handed to rvo2 per step: simulator and per-agent parameters, the
                                                    inflated radii (+ 0.01 + safety space), max speeds, preferred velocities
This script reads exactly those line ranges from /root/reference at run time and executes them against stand-ins (a Human
record, a recording rvo2 module); it stores inputs + what they produced as tests/golden/episodes_*.npz.  Nothing of the
reference's text is written to the repo.  Run in the build container:  python tests/golden/make_golden_episodes.py
"""
import os
import textwrap
import types

import numpy as np

REF = "/root/reference/crowd_sim_plus/envs"
OUT = os.path.dirname(os.path.abspath(__file__))


def ref_lines(path, lo, hi):
    with open(os.path.join(REF, path)) as f:
        lines = f.readlines()[lo - 1:hi]
    return textwrap.dedent("".join(lines))


class Human:      # crowd_sim_plus/envs/utils/agent_plus.py: the fields generate_circle_crossing_human touches
    def __init__(self, config, section, observability=None, env=None):
        self.radius, self.v_pref = config[section]["radius"], config[section]["v_pref"]
        self.px = self.py = self.gx = self.gy = None

    def set(self, px, py, gx, gy, vx, vy, theta, radius=None, v_pref=None):
        self.px, self.py, self.gx, self.gy, self.vx, self.vy, self.theta = px, py, gx, gy, vx, vy, theta


def capture_placement(tag, n_humans, seed, circle_radius, randomize, human_radius=0.20, human_v_pref=1.5,
                      robot_radius=0.25, discomfort=0.2):
    cfg = {"humans": {"radius": human_radius, "v_pref": human_v_pref}}
    robot = types.SimpleNamespace(px=0.0, py=-circle_radius, gx=0.0, gy=circle_radius, radius=robot_radius)
    self = types.SimpleNamespace(config=cfg, human_observability=None, randomize_attributes=randomize,
                                 circle_radius=circle_radius, robot=robot, humans=[],
                                 rewards={"discomfort_dist": discomfort})
    src = "def generate_circle_crossing_human(self, rng):\n" + textwrap.indent(
        ref_lines("crowd_sim_plus.py", 455, 481), "    ")
    ns = dict(np=np, norm=np.linalg.norm, Human=Human)
    exec(src, ns)
    rng = np.random.default_rng(seed)
    for _ in range(n_humans):                       # crowd_sim_plus.py:441-443
        self.humans.append(ns["generate_circle_crossing_human"](self, rng))
    np.savez(os.path.join(OUT, f"episodes_placement_{tag}.npz"), n_humans=n_humans, seed=seed, circle_radius=circle_radius,
             randomize=randomize, human_radius=human_radius, human_v_pref=human_v_pref, robot_radius=robot_radius,
             discomfort_dist=discomfort,
             pos=np.array([[h.px, h.py] for h in self.humans]), goal=np.array([[h.gx, h.gy] for h in self.humans]),
             v_pref=np.array([h.v_pref for h in self.humans]), rng_next=rng.random(4))
    print("placement", tag, len(self.humans))


def capture_square_placement(tag, n_humans, seed, square_width, randomize, circle_radius=4.0, human_radius=0.20, human_v_pref=1.5,
                             robot_radius=0.25, discomfort=0.2):
    cfg = {"humans": {"radius": human_radius, "v_pref": human_v_pref}}
    robot = types.SimpleNamespace(px=0.0, py=-circle_radius, gx=0.0, gy=circle_radius, radius=robot_radius)
    self = types.SimpleNamespace(config=cfg, human_observability=None, randomize_attributes=randomize,
                                 square_width=square_width, robot=robot, humans=[], discomfort_dist=discomfort)
    src = "def generate_square_crossing_human(self, rng):\n" + textwrap.indent(
        ref_lines("crowd_sim_plus.py", 490, 520), "    ")
    ns = dict(np=np, norm=np.linalg.norm, Human=Human)
    exec(src, ns)
    rng = np.random.default_rng(seed)
    for _ in range(n_humans):                       # crowd_sim_plus.py:437-439
        self.humans.append(ns["generate_square_crossing_human"](self, rng))
    np.savez(os.path.join(OUT, f"episodes_square_placement_{tag}.npz"), n_humans=n_humans, seed=seed, square_width=square_width,
             circle_radius=circle_radius, randomize=randomize, human_radius=human_radius, human_v_pref=human_v_pref,
             robot_radius=robot_radius, discomfort_dist=discomfort,
             pos=np.array([[h.px, h.py] for h in self.humans]), goal=np.array([[h.gx, h.gy] for h in self.humans]),
             v_pref=np.array([h.v_pref for h in self.humans]), rng_next=rng.random(4))
    print("square placement", tag, len(self.humans))


class RecordingSim:
    def __init__(self, log, *args):
        self.log = log
        log["simulator"] = args
        log["agents"], log["pref"] = [], {}

    def addAgent(self, position, *args):
        self.log["agents"].append((tuple(position),) + args)

    def setAgentPrefVelocity(self, i, v):
        self.log["pref"][i] = tuple(float(x) for x in v)

    def getNumAgents(self):
        return len(self.log["agents"])

    def doStep(self):
        pass

    def getAgentVelocity(self, i):
        return (0.0, 0.0)


def capture_orca_calls(tag, n_humans, seed, time_step):
    rng = np.random.default_rng(seed)
    log = {}
    rvo2 = types.SimpleNamespace(PyRVOSimulator=lambda *a: RecordingSim(log, *a))
    policy = types.SimpleNamespace(time_step=time_step)
    exec(ref_lines("policy/orca.py", 56, 67), dict(self=policy))                    # the parameters set by ORCA.__init__
    mk = lambda goal_far: types.SimpleNamespace(
        px=rng.uniform(-4, 4), py=rng.uniform(-4, 4), vx=rng.uniform(-1, 1), vy=rng.uniform(-1, 1),
        gx=rng.uniform(-4, 4) if goal_far else None, gy=rng.uniform(-4, 4), radius=rng.uniform(0.2, 0.4),
        v_pref=rng.uniform(0.5, 1.5))
    ego = mk(True)
    if tag.endswith("near_goal"):            # the goal closer than 1 m: the preferred velocity is NOT normalised (orca.py:115)
        ego.gx, ego.gy = ego.px + 0.3, ego.py - 0.4
    others = [mk(True) for _ in range(n_humans)]
    for a in [ego] + others:
        a.position, a.velocity = (a.px, a.py), (a.vx, a.vy)
    state = types.SimpleNamespace(self_state=ego, human_states=others)
    src = "def predict(self, state):\n" + textwrap.indent(ref_lines("policy/orca.py", 93, 129), "    ")
    ns = dict(np=np, rvo2=rvo2, ActionXY=lambda vx, vy: (vx, vy))
    exec(src, ns)
    ns["predict"](policy, state)
    agents = log["agents"]
    np.savez(os.path.join(OUT, f"episodes_orca_calls_{tag}.npz"), n_humans=n_humans, time_step=time_step,
             ego=np.array([ego.px, ego.py, ego.vx, ego.vy, ego.gx, ego.gy, ego.radius, ego.v_pref]),
             others=np.array([[o.px, o.py, o.vx, o.vy, o.radius, o.v_pref] for o in others]),
             simulator=np.array(log["simulator"], dtype=np.float64),       # time_step, neighbor_dist, max_neighbors, time_horizon, time_horizon_obst, radius, max_speed
             agent_pos=np.array([a[0] for a in agents]),
             agent_params=np.array([a[1:7] for a in agents], dtype=np.float64),   # neighbor_dist, max_neighbors, time_horizon, time_horizon_obst, radius, max_speed
             agent_vel=np.array([a[7] for a in agents]),
             pref=np.array([log["pref"][i] for i in range(len(agents))]),
             safety_space=policy.safety_space)
    print("orca calls", tag, len(agents))


if __name__ == "__main__":
    capture_placement("n5", 5, 11, 4.0, True)
    capture_placement("n25", 25, 12, 6.0, True)            # the dense crowd: many rejected draws
    capture_placement("n3_fixed_speed", 3, 13, 4.0, False)
    capture_square_placement("n5", 5, 31, 5.0, True)
    capture_square_placement("n20", 20, 32, 5.0, True)     # the shipped square_width with a dense crowd: many rejected draws
    capture_square_placement("n4_fixed_speed", 4, 33, 8.0, False)
    capture_orca_calls("n5", 5, 21, 0.25)
    capture_orca_calls("n12", 12, 22, 0.25)                # more agents than max_neighbors
    capture_orca_calls("n3_near_goal", 3, 23, 0.1)
