#!/usr/bin/env python3
"""Golden captures for the CrowdSimPlus semantics around the episode generator: static obstacles,
hallway placement with door sub-goals, wall-constrained actions, step outcomes / rewards / termination - generated from
the REFERENCE'S OWN LINES.

CrowdSimPlus cannot be imported here (gym and rvo2 are absent).  What it computes in plain Python / NumPy can still be
executed: this script reads line ranges of /root/reference/crowd_sim_plus/envs at run time, wraps them into functions and
runs them against stand-in objects (a robot / human record, a config); the reference's own helper modules
(utils/utils_plus.py, utils/action.py, utils/info_plus.py) are executed as they are.  Only inputs and what those lines
produced are stored (tests/golden/env_*.npz): nothing of the reference's text is written to the repo.

    crowd_sim_plus.py:322-421    generate_static_obstacles: wall / door segments of every rule, door geometry
    utils/human_plus.py:19-79    Human.get_g_xy / set_g_xy / set: the door sub-goal of the hallway_static* / bottleneck rules
    crowd_sim_plus.py:522-607    generate_hallway_human: placement with its draw order, rejection against robot, humans, walls
    crowd_sim_plus.py:869-989    constrain_agent_action_exact: an action cut short at walls and wall ends
    crowd_sim_plus.py:1067-1166  step(): collision / frozen / goal / timeout detection and the reward terms
    crowd_sim_plus.py:1203-1206  step(): the first time a human reaches its (current) goal; utils/agent_plus.py:217 reached_destination

Run in the build container:  python tests/golden/make_golden_env.py
"""
import os
import textwrap
import types

import numpy as np

REF = "/root/reference/crowd_sim_plus/envs"
OUT = os.path.dirname(os.path.abspath(__file__))
RULES = ["hallway", "hallway_static", "hallway_static_with_back", "hallway_bottleneck", "hallway_squeeze", "rectangle", "left_wall"]


def ref_lines(path, lo, hi):
    with open(os.path.join(REF, path)) as f:
        lines = f.readlines()[lo - 1:hi]
    return textwrap.dedent("".join(lines))


def ref_namespace():
    """The reference's helper modules, executed (not copied): geometry helpers, action tuples, info records."""
    ns = dict(np=np, norm=np.linalg.norm)
    for mod in ("utils/utils_plus.py", "utils/action.py", "utils/info_plus.py"):
        exec(compile(open(os.path.join(REF, mod)).read(), mod, "exec"), ns)
    from copy import deepcopy
    ns["deepcopy"] = deepcopy
    return ns


def make_functions():
    ns = ref_namespace()
    # Human: the reference's door-sub-goal methods on a bare record (the real class pulls in the policy factory)
    human_src = ("class Human:\n"
                 "    def __init__(self, config, section, fully_observable=False, env=None):\n"
                 "        self.radius = config.getfloat(section, 'radius')\n"
                 "        self.v_pref = config.getfloat(section, 'v_pref')\n"
                 "        self.env = env\n"
                 "        self.px = self.py = self.gx = self.gy = None\n"
                 + textwrap.indent(ref_lines("utils/human_plus.py", 19, 79), "    "))
    exec(human_src, ns)
    exec("def generate_static_obstacles(self, rule, static_obstacles=None):\n"
         + textwrap.indent(ref_lines("crowd_sim_plus.py", 329, 421), "    "), ns)
    exec("def generate_hallway_human(self, rng):\n" + textwrap.indent(ref_lines("crowd_sim_plus.py", 523, 607), "    "), ns)
    exec("def constrain_agent_action_exact(self, agent, action):\n"
         + textwrap.indent(ref_lines("crowd_sim_plus.py", 876, 989), "    "), ns)
    exec("def step_outcome(self, action, human_actions, stat_collision, update):\n"
         + textwrap.indent(ref_lines("crowd_sim_plus.py", 1067, 1166), "    ")
         + "\n    return reward, done, info, dmin, collision, frozen_robot, reached_goal, curr_dist_to_goal\n", ns)
    return ns


class Config:
    def __init__(self, human_radius=0.20, human_v_pref=1.5, policy="orca_plus"):
        self.d = {("humans", "radius"): human_radius, ("humans", "v_pref"): human_v_pref, ("humans", "policy"): policy}

    def get(self, s, k):
        return self.d[(s, k)]

    def getfloat(self, s, k):
        return float(self.d[(s, k)])


class Agent:
    """position / goal / radius record with the holonomic kinematics of utils/agent_plus.py:175-185"""
    kinematics = "holonomic"

    def __init__(self, px, py, gx, gy, radius, time_step=0.25):
        self.px, self.py, self.gx, self.gy, self.radius, self.time_step = px, py, gx, gy, radius, time_step

    def compute_position(self, action, dt):
        return self.px + action.vx * dt, self.py + action.vy * dt

    def get_goal_position(self):
        return self.gx, self.gy


def make_env(ns, rule, circle_radius=1.0, rect_width=1.75, rect_height=4.0, robot_radius=0.25, discomfort=0.2,
             randomize=True, cfg=None):
    env = types.SimpleNamespace(circle_radius=circle_radius, rect_width=rect_width, rect_height=rect_height,
                                robot=Agent(0.0, -circle_radius, 0.0, circle_radius, robot_radius),
                                config=cfg or Config(), human_observability=None, randomize_attributes=randomize,
                                rewards={"discomfort_dist": discomfort}, humans=[], sim_env=rule, static_obstacles=[])
    ns["generate_static_obstacles"](env, rule)
    return env


def door_fields(env):
    keys = ("door_y_max", "door_y_min", "door_x_mid", "door_y_mid_max", "door_y_mid_min", "door_width")
    return {k: float(getattr(env, k, np.nan)) for k in keys}


def capture_static_obstacles(ns):
    out = {}
    for tag, geo in {"shipped": dict(circle_radius=1.0, rect_width=1.75, rect_height=4.0, robot_radius=0.25),
                     "wide": dict(circle_radius=4.0, rect_width=6.0, rect_height=8.0, robot_radius=0.3)}.items():
        for rule in RULES:
            env = make_env(ns, rule, **geo)
            out[f"{tag}__{rule}__segments"] = np.array(env.static_obstacles, dtype=np.float64).reshape(-1, 2, 2)
            out[f"{tag}__{rule}__doors"] = np.array(list(door_fields(env).values()))
        out[f"{tag}__geometry"] = np.array([geo["circle_radius"], geo["rect_width"], geo["rect_height"], geo["robot_radius"]])
    np.savez(os.path.join(OUT, "env_static_obstacles.npz"), **out)
    print("static obstacles", len(out))


def capture_hallway_placement(ns, tag, rule, n_humans, seed, randomize=True, **geo):
    env = make_env(ns, rule, randomize=randomize, **geo)
    rng = np.random.default_rng(seed)
    for _ in range(n_humans):                       # crowd_sim_plus.py:444-447
        env.humans.append(ns["generate_hallway_human"](env, rng))
    hs = env.humans
    np.savez(os.path.join(OUT, f"env_hallway_placement_{tag}.npz"), rule=rule, n_humans=n_humans, seed=seed, randomize=randomize,
             geometry=np.array([env.circle_radius, env.rect_width, env.rect_height, env.robot.radius]),
             human_radius=hs[0].radius, human_v_pref=env.config.getfloat("humans", "v_pref"),
             discomfort_dist=env.rewards["discomfort_dist"],
             pos=np.array([[h.px, h.py] for h in hs]), final_goal=np.array([[h.final_gx, h.final_gy] for h in hs]),
             goal=np.array([[h.gx, h.gy] for h in hs]),            # after set(): the door sub-goal where one applies
             v_pref=np.array([h.v_pref for h in hs]), theta=np.array([h.theta for h in hs]), rng_next=rng.random(4))
    print("hallway placement", tag, rule, n_humans)


def capture_door_subgoals(ns):
    """Human.get_g_xy over a grid of positions and goals, every rule."""
    rng = np.random.default_rng(3)
    out = {}
    for rule in RULES:
        env = make_env(ns, rule)
        h = ns["Human"](env.config, "humans", env=env)
        pts, goals, got = [], [], []
        for _ in range(400):
            px, py = rng.uniform(-0.9, 0.9), rng.uniform(-4.0, 4.0)
            gx, gy = rng.uniform(-0.9, 0.9), rng.uniform(-4.0, 4.0)
            if rng.random() < 0.2:
                px, py = rng.normal(0.0, 0.3), rng.normal(0.0, 0.3)     # around the door centre: the switch back to the final goal
            h.final_gx, h.final_gy = gx, gy
            pts.append((px, py)); goals.append((gx, gy)); got.append(h.get_g_xy(px, py))
        out[f"{rule}__pos"], out[f"{rule}__final_goal"], out[f"{rule}__goal"] = np.array(pts), np.array(goals), np.array(got)
    np.savez(os.path.join(OUT, "env_door_subgoals.npz"), **out)
    print("door sub-goals", len(out))


def capture_constrained_actions(ns, tag, rule, n, seed, radius=0.21, time_step=0.25, **geo):
    """constrain_agent_action_exact for n random (position, action) pairs; a share of them placed within a step of a wall or a
    wall end so that every branch runs (free, mid-segment, end point, already touching, heading straight at an end)."""
    env = make_env(ns, rule, **geo)
    segs = np.array(env.static_obstacles, dtype=np.float64).reshape(-1, 2, 2)
    rng = np.random.default_rng(seed)
    A = ns["ActionXY"]
    pos, act, got = [], [], []
    while len(pos) < n:
        kind = rng.integers(0, 5)
        s = segs[rng.integers(0, len(segs))]
        d = s[1] - s[0]
        nrm = np.array([-d[1], d[0]]) / np.linalg.norm(d)
        if kind == 0:          # anywhere
            p = np.array([rng.uniform(-env.rect_width, env.rect_width), rng.uniform(-env.rect_height, env.rect_height)])
            v = rng.uniform(-1.5, 1.5, 2)
        elif kind == 1:        # next to the middle of a wall, moving at it
            p = s[0] + rng.uniform(0.1, 0.9) * d + nrm * rng.choice([-1, 1]) * rng.uniform(radius + 1e-6, radius + 0.3)
            v = rng.uniform(-1.5, 1.5, 2)
        elif kind == 2:        # next to a wall end
            e = s[rng.integers(0, 2)]
            ang = rng.uniform(0, 2 * np.pi)
            p = e + rng.uniform(radius + 1e-6, radius + 0.35) * np.array([np.cos(ang), np.sin(ang)])
            v = rng.uniform(-1.5, 1.5, 2)
        elif kind == 3:        # heading straight at a wall end
            e = s[rng.integers(0, 2)]
            ang = rng.uniform(0, 2 * np.pi)
            p = e + rng.uniform(radius + 0.01, radius + 0.3) * np.array([np.cos(ang), np.sin(ang)])
            v = (e - p) / np.linalg.norm(e - p) * rng.uniform(0.3, 1.5)
        else:                  # touching a wall already (what the previous constrained step leaves behind: radius + 1e-7)
            p = s[0] + rng.uniform(0.1, 0.9) * d + nrm * rng.choice([-1, 1]) * (radius + 1e-7)
            v = rng.uniform(-1.5, 1.5, 2)
        # the reference never starts inside a wall: skip positions closer than the radius to any segment
        if min(ns["point_to_segment_dist"](q[0][0], q[0][1], q[1][0], q[1][1], p[0], p[1]) for q in segs) < radius:
            continue
        agent = Agent(float(p[0]), float(p[1]), 0.0, 0.0, radius, time_step)
        try:
            c = ns["constrain_agent_action_exact"](env, agent, A(float(v[0]), float(v[1])))
        except AssertionError:         # (the reference asserts on its own degenerate triangles: not a case to pin)
            continue
        pos.append(p); act.append(v); got.append((c.vx, c.vy))
    np.savez(os.path.join(OUT, f"env_constrain_{tag}.npz"), rule=rule, radius=radius, time_step=time_step, segments=segs,
             geometry=np.array([env.circle_radius, env.rect_width, env.rect_height, env.robot.radius]),
             pos=np.array(pos), action=np.array(act), constrained=np.array(got))
    changed = int((np.abs(np.array(got) - np.array(act)).max(axis=1) > 0).sum())
    print("constrained actions", tag, rule, n, "changed:", changed)


def capture_constrained_rot_actions(ns, tag, rule, n, seed, radius=0.25, time_step=0.25, **geo):
    """The same for a NON-holonomic agent (the MPC's unicycle robot: ActionRot(v, r), position by utils/agent_plus.py:175-185 executed from
    its lines): the constrained action keeps the rotation and shortens |v| (crowd_sim_plus.py:976-987), forwards and backwards."""
    env = make_env(ns, rule, **geo)
    segs = np.array(env.static_obstacles, dtype=np.float64).reshape(-1, 2, 2)
    rng = np.random.default_rng(seed)
    R = ns["ActionRot"]
    cls_ns = dict(np=np)
    exec("class Uni:\n    kinematics = 'unicycle'\n    def check_validity(self, action):\n        pass\n"
         + textwrap.indent(ref_lines("utils/agent_plus.py", 175, 185), "    "), cls_ns)
    pos, theta, act, got = [], [], [], []
    while len(pos) < n:
        s = segs[rng.integers(0, len(segs))]
        d = s[1] - s[0]
        nrm = np.array([-d[1], d[0]]) / np.linalg.norm(d)
        kind = rng.integers(0, 4)
        if kind == 0:
            p = np.array([rng.uniform(-env.rect_width, env.rect_width), rng.uniform(-env.rect_height, env.rect_height)])
        elif kind == 1:
            p = s[0] + rng.uniform(0.1, 0.9) * d + nrm * rng.choice([-1, 1]) * rng.uniform(radius + 1e-6, radius + 0.3)
        elif kind == 2:
            e = s[rng.integers(0, 2)]
            ang = rng.uniform(0, 2 * np.pi)
            p = e + rng.uniform(radius + 1e-6, radius + 0.35) * np.array([np.cos(ang), np.sin(ang)])
        else:
            p = s[0] + rng.uniform(0.1, 0.9) * d + nrm * rng.choice([-1, 1]) * (radius + 1e-7)
        if min(ns["point_to_segment_dist"](q[0][0], q[0][1], q[1][0], q[1][1], p[0], p[1]) for q in segs) < radius:
            continue
        th, v, r = rng.uniform(-np.pi, np.pi), rng.uniform(-1.0, 1.5), rng.uniform(-0.6, 0.6)
        if rng.random() < 0.05:
            v = 0.0
        agent = cls_ns["Uni"]()
        agent.px, agent.py, agent.theta, agent.radius, agent.time_step = float(p[0]), float(p[1]), float(th), radius, time_step
        try:
            c = ns["constrain_agent_action_exact"](env, agent, R(float(v), float(r)))
        except AssertionError:
            continue
        pos.append(p); theta.append(th); act.append((v, r)); got.append((c.v, c.r))
    np.savez(os.path.join(OUT, f"env_rotconstrain_{tag}.npz"), rule=rule, radius=radius, time_step=time_step, segments=segs,
             pos=np.array(pos), theta=np.array(theta), action=np.array(act), constrained=np.array(got))
    print("constrained unicycle actions", tag, rule, n, "changed:", int((np.array(got)[:, 0] != np.array(act)[:, 0]).sum()))


INFO_KEYS = ("ReachGoal", "Timeout", "Collision", "WallCollision", "Frozen", "Danger", "Progress", "AngularSmoothness", "LinearSmoothness")


def capture_step_outcomes(ns, tag, n, seed, rewards, detailed, unicycle=False):
    """The outcome block of step() for n random situations of one robot and three humans."""
    rng = np.random.default_rng(seed)
    A = ns["ActionXY"]
    uni_ns = dict(np=np)      # the unicycle robot's position update: utils/agent_plus.py:175-185 from its lines
    exec("class Uni:\n    kinematics = 'unicycle'\n    def check_validity(self, action):\n        pass\n"
         "    def get_goal_position(self):\n        return self.gx, self.gy\n"
         + textwrap.indent(ref_lines("utils/agent_plus.py", 175, 185), "    "), uni_ns)
    rec = {k: [] for k in ("robot_theta", "robot_rot", "robot", "robot_action", "humans", "human_actions", "human_radius", "global_time", "stat_collision",
                           "prev_dist", "prev_angular", "prev_linear", "reward", "done", "dmin", "collision", "frozen", "reached",
                           "curr_dist", "info_vals", "info_present", "next_prev_dist", "next_prev_angular", "next_prev_linear")}
    time_limit, time_step = 30.0, 0.25
    for i in range(n):
        kind = rng.integers(0, 6)
        goal = np.array([0.0, 1.0])
        p = rng.uniform(-1.5, 1.5, 2)
        v = rng.uniform(-1.0, 1.0, 2)
        if kind == 1:      # about to reach the goal
            p = goal - v * time_step + rng.normal(0, 0.1, 2)
        if kind == 2:      # frozen
            v = rng.normal(0, 0.02, 2)
        hp = rng.uniform(-2.0, 2.0, (3, 2))
        hv = rng.uniform(-1.0, 1.0, (3, 2))
        hr = rng.uniform(0.18, 0.3, 3)
        if kind == 3:      # a human ends the step on top of the robot
            hp[1] = p + v * time_step - hv[1] * time_step + rng.normal(0, 0.15, 2)
        if kind == 4:      # inside the discomfort distance
            ang = rng.uniform(0, 2 * np.pi)
            hp[2] = p + v * time_step - hv[2] * time_step + (0.25 + hr[2] + rng.uniform(0.0, 0.25)) * np.array([np.cos(ang), np.sin(ang)])
        gt = float(rng.choice([rng.uniform(0, 29.0), 30.0, 30.25])) if kind == 5 else float(rng.uniform(0, 29.0))
        robot = Agent(float(p[0]), float(p[1]), float(goal[0]), float(goal[1]), 0.25, time_step)
        th = vr = vs = 0.0
        if unicycle:      # the same world-frame step, expressed as ActionRot(v, r) on heading th: th + r = the direction of v (or its opposite)
            vr = float(rng.uniform(-0.6, 0.6))
            vs = float(rng.choice([-1.0, 1.0]) * np.linalg.norm(v))
            th = float(np.arctan2(v[1], v[0]) - vr + (np.pi if vs < 0 else 0.0))
            robot = uni_ns["Uni"]()
            robot.px, robot.py, robot.gx, robot.gy, robot.radius, robot.time_step, robot.theta = (
                float(p[0]), float(p[1]), float(goal[0]), float(goal[1]), 0.25, time_step, th)
        humans = [Agent(float(hp[j, 0]), float(hp[j, 1]), 0.0, 0.0, float(hr[j]), time_step) for j in range(3)]
        first = rng.random() < 0.2
        env = types.SimpleNamespace(robot=robot, humans=humans, time_step=time_step, global_time=gt, time_limit=time_limit,
                                    rewards=dict(rewards), detailed_reward_return=detailed, robot_goal_pos=goal.copy(),
                                    robot_prev_dist_to_goal=float(np.linalg.norm(goal - p) + rng.normal(0, 0.05)),
                                    prev_action_angular=None if first else float(rng.uniform(-np.pi, np.pi)),
                                    prev_action_linear=None if first else float(rng.uniform(0, 1.0)))
        stat = bool(rng.random() < 0.15)
        pd, pa, pl = env.robot_prev_dist_to_goal, env.prev_action_angular, env.prev_action_linear
        r, done, info, dmin, coll, frozen, reached, cd = ns["step_outcome"](
            env, ns["ActionRot"](vs, vr) if unicycle else A(float(v[0]), float(v[1])),
            [A(float(hv[j, 0]), float(hv[j, 1])) for j in range(3)], stat, True)
        rec["robot_theta"].append(th); rec["robot_rot"].append([vs, vr])
        rec["robot"].append([p[0], p[1], goal[0], goal[1], 0.25]); rec["robot_action"].append(v)
        rec["humans"].append(hp); rec["human_actions"].append(hv); rec["human_radius"].append(hr)
        rec["global_time"].append(gt); rec["stat_collision"].append(stat)
        rec["prev_dist"].append(pd); rec["prev_angular"].append(np.nan if pa is None else pa)
        rec["prev_linear"].append(np.nan if pl is None else pl)
        rec["reward"].append(r); rec["done"].append(done); rec["dmin"].append(dmin); rec["collision"].append(coll)
        rec["frozen"].append(bool(frozen)); rec["reached"].append(bool(reached)); rec["curr_dist"].append(cd)
        rec["info_vals"].append([float(info[k].val) if k in info else 0.0 for k in INFO_KEYS])
        rec["info_present"].append([k in info and (k in ("Progress",) or float(info[k].val) != 0.0 or k in ("AngularSmoothness", "LinearSmoothness")) for k in INFO_KEYS])
        rec["next_prev_dist"].append(env.robot_prev_dist_to_goal)
        rec["next_prev_angular"].append(np.nan if env.prev_action_angular is None else env.prev_action_angular)
        rec["next_prev_linear"].append(np.nan if env.prev_action_linear is None else env.prev_action_linear)
    np.savez(os.path.join(OUT, f"env_step_outcomes_{tag}.npz"), time_limit=time_limit, time_step=time_step, detailed=detailed,
             unicycle=unicycle, reward_keys=np.array(sorted(rewards)), reward_vals=np.array([float(rewards[k]) for k in sorted(rewards)]),
             info_keys=np.array(INFO_KEYS), **{k: np.array(v) for k, v in rec.items()})
    print("step outcomes", tag, n, "done:", int(np.sum(rec["done"])), "collisions:", int(np.sum(rec["collision"])),
          "reached:", int(np.sum(rec["reached"])), "frozen:", int(np.sum(rec["frozen"])))


class RecordingSim:
    """stand-in for rvo2.PyRVOSimulator: records what ORCAPlus.predict hands to it"""
    def __init__(self, log, *args):
        self.log = log
        log["simulator"] = args
        log["agents"], log["pref"], log["obstacles"], log["processed"] = [], {}, [], 0

    def addObstacle(self, line):
        self.log["obstacles"].append([tuple(line[0]), tuple(line[1])])

    def processObstacles(self):
        self.log["processed"] += 1

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


def capture_orca_plus_calls(ns, tag, rule, n_humans, seed, time_step=0.25, near_goal=False):
    """What ORCAPlus.predict (policy/orca_plus.py:43-84) hands to rvo2 for one human's step: the walls, the simulator and agent
    parameters (radius + 0.01 + the humans' safety space), the preferred velocity (capped at v_pref - 1e-3)."""
    rng = np.random.default_rng(seed)
    log = {}
    rvo2 = types.SimpleNamespace(PyRVOSimulator=lambda *a: RecordingSim(log, *a))
    policy = types.SimpleNamespace(time_step=time_step, sim=None)
    exec(ref_lines("policy/orca.py", 56, 67), dict(self=policy))                    # ORCA.__init__'s parameters
    # ORCAPlus.configure(config, 'humans') (orca_plus.py:15-27, called from Human.__init__): radius and safety space of [humans]
    cfg = types.SimpleNamespace(getfloat=lambda s, k: {("env", "time_step"): time_step, ("humans", "radius"): 0.20,
                                                       ("humans", "safety_space"): 0.05}[(s, k)])
    src = "def configure(self, config, section='orca_plus'):\n" + textwrap.indent(ref_lines("policy/orca_plus.py", 16, 27), "    ")
    cns = dict(logging=__import__("logging"))
    exec(src, cns)
    cns["configure"](policy, cfg, "humans")
    env = make_env(ns, rule)
    mk = lambda: types.SimpleNamespace(px=rng.uniform(-0.6, 0.6), py=rng.uniform(-3, 3), vx=rng.uniform(-1, 1), vy=rng.uniform(-1, 1),
                                       gx=rng.uniform(-0.6, 0.6), gy=rng.uniform(-3, 3), radius=0.20, v_pref=rng.uniform(0.5, 1.5))
    ego = mk()
    if near_goal:
        ego.gx, ego.gy = ego.px + 0.2, ego.py - 0.3
    others = [mk() for _ in range(n_humans)]
    for a in [ego] + others:
        a.position, a.velocity = (a.px, a.py), (a.vx, a.vy)
    state = types.SimpleNamespace(self_state=ego, human_states=others, static_obs=[[tuple(q[0]), tuple(q[1])] for q in env.static_obstacles])
    src = "def predict(self, state):\n" + textwrap.indent(ref_lines("policy/orca_plus.py", 43, 84), "    ")
    pns = dict(np=np, rvo2=rvo2, ActionXY=lambda vx, vy: (vx, vy))
    exec(src, pns)
    pns["predict"](policy, state)
    agents = log["agents"]
    np.savez(os.path.join(OUT, f"env_orca_plus_calls_{tag}.npz"), rule=rule, n_humans=n_humans, time_step=time_step,
             ego=np.array([ego.px, ego.py, ego.vx, ego.vy, ego.gx, ego.gy, ego.radius, ego.v_pref]),
             others=np.array([[o.px, o.py, o.vx, o.vy, o.radius, o.v_pref] for o in others]),
             simulator=np.array(log["simulator"], dtype=np.float64), obstacles=np.array(log["obstacles"], dtype=np.float64).reshape(-1, 2, 2),
             processed=log["processed"], agent_pos=np.array([a[0] for a in agents]),
             agent_params=np.array([a[1:7] for a in agents], dtype=np.float64), agent_vel=np.array([a[7] for a in agents]),
             pref=np.array([log["pref"][i] for i in range(len(agents))]), safety_space=policy.safety_space, policy_radius=policy.radius)
    print("orca_plus calls", tag, rule, len(agents), len(log["obstacles"]))


SFM_PARAMS = dict(radius=0.20, A=3.0, B=0.18, KI=1.0, A_static=2.0, B_static=0.025, A_bottleneck=6.0, B_bottleneck=0.12)   # env.config [humans]


def sfm_predict(ns):
    """SFM.predict (policy/social_force.py:39-95), the reference's lines as a function of (policy record, state)."""
    if "sfm_predict" not in ns:
        exec("def sfm_predict(self, state):\n" + textwrap.indent(ref_lines("policy/social_force.py", 39, 95), "    "), ns)
    return ns["sfm_predict"]


def capture_sfm_calls(ns, tag, rule, n, seed, n_others=3, time_step=0.25):
    """n single calls of SFM.predict: one human among n_others agents (the last of them the robot, radius 0.25) and the walls of the rule."""
    env = make_env(ns, rule)
    rng = np.random.default_rng(seed)
    pol = types.SimpleNamespace(time_step=time_step, is_bottleneck=(rule == "hallway_bottleneck"), **SFM_PARAMS)
    predict = sfm_predict(ns)
    rec = {k: [] for k in ("ego", "others", "action")}
    for _ in range(n):
        mk = lambda r: types.SimpleNamespace(px=rng.uniform(-0.6, 0.6), py=rng.uniform(-2.5, 2.5), vx=rng.uniform(-1, 1), vy=rng.uniform(-1, 1),
                                             gx=rng.uniform(-0.6, 0.6), gy=rng.uniform(-3, 3), radius=r, v_pref=rng.uniform(0.5, 1.5))
        ego = mk(0.20)
        others = [mk(0.20) for _ in range(n_others - 1)] + [mk(0.25)]
        if rng.random() < 0.1:
            ego.gx, ego.gy = ego.px, ego.py                                  # standing on its goal: the 1e-6 guard
        state = types.SimpleNamespace(self_state=ego, human_states=others, static_obs=[[tuple(q[0]), tuple(q[1])] for q in env.static_obstacles])
        a = predict(pol, state)
        rec["ego"].append([ego.px, ego.py, ego.vx, ego.vy, ego.gx, ego.gy, ego.radius, ego.v_pref])
        rec["others"].append([[o.px, o.py, o.vx, o.vy, o.radius] for o in others])
        rec["action"].append([a.vx, a.vy])
    np.savez(os.path.join(OUT, f"env_sfm_calls_{tag}.npz"), rule=rule, time_step=time_step, params=np.array(list(SFM_PARAMS.values())),
             param_keys=np.array(list(SFM_PARAMS)), **{k: np.array(v) for k, v in rec.items()})
    print("sfm calls", tag, rule, n)


def capture_sfm_rollout(ns, tag, rule, n_humans, seed, steps, starts_moving=10, time_step=0.25, unicycle=False):
    """A whole episode with social-force humans, step by step with the reference's own lines: placement (generate_hallway_human),
    per step every human's observation (the other humans, then the robot: crowd_sim_plus.py:1044-1052), SFM.predict, the wall
    constraint, the outcome block, the position update and the door sub-goal (Agent.step / Human.step).  The robot stands for the
    first `starts_moving` steps (the dummy start, :707-721) and then heads for its goal at its preferred speed (scripted: the
    reference's robot is the MPC)."""
    env = make_env(ns, rule, cfg=Config(policy="sfm"))
    env.time_step, env.global_time, env.time_limit, env.detailed_reward_return = time_step, -starts_moving * time_step, 30.0, False
    env.rewards = {"success_reward": 1.0, "collision_penalty": -0.25, "freezing_penalty": -0.125, "discomfort_dist": 0.2,
                   "discomfort_penalty_factor": 0.5, "discomfort": True, "timeout": -1.0, "wall_collision_penalty": -1.0}
    rng = np.random.default_rng(seed)
    for _ in range(n_humans):
        env.humans.append(ns["generate_hallway_human"](env, rng))
    A = ns["ActionXY"]
    for h in env.humans:
        h.time_step, h.vx, h.vy, h.kinematics = time_step, 0.0, 0.0, "holonomic"
        h.compute_position = (lambda self: (lambda action, dt: (self.px + action.vx * dt, self.py + action.vy * dt)))(h)
    robot = env.robot
    if unicycle:      # the MPC's robot (sicnav_acados.py:143): ActionRot commands, position and heading by utils/agent_plus.py:175-214 from its lines
        uni_ns = dict(np=np)
        exec("class Uni:\n    kinematics = 'unicycle'\n    def check_validity(self, action):\n        pass\n"
             + textwrap.indent(ref_lines("utils/agent_plus.py", 175, 185), "    ") + "\n"
             + textwrap.indent(ref_lines("utils/agent_plus.py", 199, 214), "    "), uni_ns)
        old, robot = robot, uni_ns["Uni"]()
        robot.px, robot.py, robot.gx, robot.gy, robot.radius, robot.time_step, robot.theta = old.px, old.py, old.gx, old.gy, old.radius, time_step, np.pi / 2
        robot.get_goal_position = lambda: (robot.gx, robot.gy)
        env.robot = robot
    robot.vx = robot.vy = 0.0
    robot.v_pref = 1.0
    env.robot_goal_pos = np.array([robot.gx, robot.gy])
    env.robot_prev_dist_to_goal = 0
    env.prev_action_angular = env.prev_action_linear = None
    pol = types.SimpleNamespace(time_step=time_step, is_bottleneck=(rule == "hallway_bottleneck"), **SFM_PARAMS)
    predict = sfm_predict(ns)
    obs = lambda a: types.SimpleNamespace(px=a.px, py=a.py, vx=a.vx, vy=a.vy, radius=a.radius)
    full = lambda a: types.SimpleNamespace(px=a.px, py=a.py, vx=a.vx, vy=a.vy, radius=a.radius, gx=a.gx, gy=a.gy, v_pref=a.v_pref)
    start = dict(pos=np.array([[h.px, h.py] for h in env.humans]), final_goal=np.array([[h.final_gx, h.final_gy] for h in env.humans]),
                 v_pref=np.array([h.v_pref for h in env.humans]))
    # the human_times bookkeeping of step() (crowd_sim_plus.py:1203-1206) and Agent.reached_destination (utils/agent_plus.py:217), both
    # executed from the reference's lines; the stand-in humans answer get_position / get_goal_position like agent_plus.py:136-144
    env.human_times = [0] * n_humans
    reached_src = "def reached_destination(self):\n" + textwrap.indent(ref_lines("utils/agent_plus.py", 217, 217), "    ")
    times_src = "def record_human_times(self):\n" + textwrap.indent(ref_lines("crowd_sim_plus.py", 1203, 1206), "    ")
    tns = dict(np=np, norm=np.linalg.norm)
    exec(reached_src, tns)
    exec(times_src, tns)
    for h in env.humans:
        h.get_position = (lambda self: (lambda: (self.px, self.py)))(h)
        h.get_goal_position = (lambda self: (lambda: (self.gx, self.gy)))(h)
        h.reached_destination = (lambda self: (lambda: tns["reached_destination"](self)))(h)
    traj, racts, outs, thetas = [], [], [], []
    for s in range(starts_moving + steps):
        human_actions = []
        for h in env.humans:                                                            # crowd_sim_plus.py:1044-1056
            ob = [obs(o) for o in env.humans if o is not h] + [obs(robot)]
            state = types.SimpleNamespace(self_state=full(h), human_states=ob, static_obs=env.static_obstacles)
            human_actions.append(ns["constrain_agent_action_exact"](env, h, predict(pol, state)))
        if unicycle:
            if s < starts_moving:
                want = ns["ActionRot"](0.0, 0.0)                                            # crowd_sim_plus.py:718
            else:      # scripted: turn towards a point that swings across the corridor (so that walls are met), at most 0.5 rad per step
                k = s - starts_moving
                tx, ty = robot.gx + 1.2 * np.sin(0.45 * k), robot.gy
                err = (np.arctan2(ty - robot.py, tx - robot.px) - robot.theta + np.pi) % (2 * np.pi) - np.pi
                want = ns["ActionRot"](robot.v_pref if k % 11 != 7 else -0.4, float(np.clip(err, -0.5, 0.5)))
            act = ns["constrain_agent_action_exact"](env, robot, want)
            stat = act.v != want.v
        else:
            if s < starts_moving:
                want = A(0.0, 0.0)
            else:
                d = np.array([robot.gx - robot.px, robot.gy - robot.py])
                want = A(*(d / max(np.linalg.norm(d), 1e-9) * robot.v_pref))
            act = ns["constrain_agent_action_exact"](env, robot, want)
            stat = act.vx != want.vx
        r, done, info, dmin, coll, frozen, reached, cd = ns["step_outcome"](env, act, human_actions, stat, True)
        outs.append([r, float(done), dmin, float(coll), float(reached)])
        if unicycle:
            racts.append([want.v, want.r])
            robot.step(act)                                                             # Agent.step, from the reference's lines
            thetas.append(robot.theta)
        else:
            racts.append([want.vx, want.vy])
            robot.px, robot.py = robot.compute_position(act, time_step)                 # Agent.step
            robot.vx, robot.vy = act.vx, act.vy
        for h, a in zip(env.humans, human_actions):                                     # Human.step
            h.px, h.py = h.px + a.vx * time_step, h.py + a.vy * time_step
            h.vx, h.vy = a.vx, a.vy
            h.set_g_xy(h.px, h.py)
        env.global_time += time_step
        tns["record_human_times"](env)
        traj.append([[robot.px, robot.py]] + [[h.px, h.py] for h in env.humans])
    np.savez(os.path.join(OUT, f"env_rollout_sfm_{tag}.npz"), rule=rule, n_humans=n_humans, seed=seed, steps=steps, starts_moving=starts_moving,
             time_step=time_step, traj=np.array(traj), robot_wanted=np.array(racts), outcomes=np.array(outs),
             human_times=np.array(env.human_times, dtype=np.float64), unicycle=unicycle, robot_theta=np.array(thetas), **start)
    print("sfm rollout", tag, rule, n_humans, "steps", steps, "robot end", traj[-1][0])


if __name__ == "__main__":
    ns = make_functions()
    capture_sfm_calls(ns, "hallway", "hallway", 300, 71)
    capture_sfm_calls(ns, "bottleneck", "hallway_bottleneck", 300, 72, n_others=4)
    capture_sfm_rollout(ns, "hallway_n3", "hallway", 3, 81, 30)
    capture_sfm_rollout(ns, "static_n4", "hallway_static", 4, 82, 40)
    capture_sfm_rollout(ns, "bottleneck_n3", "hallway_bottleneck", 3, 83, 30)
    capture_sfm_rollout(ns, "hallway_n4_long", "hallway", 4, 84, 70)      # long enough for humans to arrive: human_times
    capture_sfm_rollout(ns, "unicycle_hallway_n3", "hallway", 3, 85, 40, unicycle=True)
    capture_sfm_rollout(ns, "unicycle_static_n3", "hallway_static", 3, 86, 40, unicycle=True)
    capture_orca_plus_calls(ns, "hallway_n3", "hallway", 3, 61)
    capture_orca_plus_calls(ns, "static_n5", "hallway_static", 5, 62)
    capture_orca_plus_calls(ns, "near_goal", "hallway_bottleneck", 2, 63, near_goal=True)
    capture_static_obstacles(ns)
    capture_door_subgoals(ns)
    capture_hallway_placement(ns, "shipped_n3", "hallway", 3, 31)                       # env.config: hallway, 3 humans
    capture_hallway_placement(ns, "static_n5", "hallway_static", 5, 32)                 # door sub-goals enter the goal test
    capture_hallway_placement(ns, "bottleneck_n8", "hallway_bottleneck", 8, 33)         # crowded: the effective height grows
    capture_hallway_placement(ns, "squeeze_n4_fixed", "hallway_squeeze", 4, 34, randomize=False)
    capture_constrained_actions(ns, "hallway", "hallway", 300, 41)
    capture_constrained_actions(ns, "static", "hallway_static", 500, 42)
    capture_constrained_actions(ns, "squeeze", "hallway_squeeze", 300, 43)
    capture_constrained_actions(ns, "rectangle", "rectangle", 200, 44, circle_radius=4.0, rect_width=6.0, rect_height=8.0)
    capture_constrained_rot_actions(ns, "hallway", "hallway", 300, 45)
    capture_constrained_rot_actions(ns, "static", "hallway_static", 400, 46)
    # the shipped [reward] section (env.config:66-71) as configure() leaves it for non-RL testing (crowd_sim_plus.py:110-128)
    shipped = {"success_reward": 1.0, "collision_penalty": -0.25, "freezing_penalty": -0.125, "discomfort_dist": 0.2,
               "discomfort_penalty_factor": 0.5, "discomfort": True, "timeout": -1.0, "wall_collision_penalty": -1.0}
    capture_step_outcomes(ns, "shipped", 400, 51, shipped, False)
    full = dict(shipped, progress_factor=0.1, angular_smoothness_factor=-0.01, linear_smoothness_factor=-0.02)
    capture_step_outcomes(ns, "all_terms", 400, 52, full, True)
    capture_step_outcomes(ns, "unicycle_all_terms", 400, 53, full, True, unicycle=True)
    capture_step_outcomes(ns, "unicycle_shipped", 300, 54, shipped, False, unicycle=True)
