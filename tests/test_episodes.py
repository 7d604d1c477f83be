"""Batched episode generator: the vectorised ORCA of safe-interactive-crowdnav_amd/episodes.py
against the scalar restatement of the published algorithm (oracle/orca_oracle.py), against a brute-force search of
the velocity disc, and through properties of the generated crowds.  PARITY UNPINNED: rvo2, which the reference calls
(crowd_sim_plus/envs/policy/orca.py:82-133), is absent here - see the headers of both files."""
import numpy as np
import pytest

from oracle import orca_oracle as OO
from safe_interactive_crowdnav_amd import episodes as EP
from safe_interactive_crowdnav_amd import scene as SC


def _random_crowd(rng, E, n, spread, speed):
    pos = rng.uniform(-spread, spread, (E, n, 2))
    vel = rng.uniform(-speed, speed, (E, n, 2))
    rad = rng.uniform(0.2, 0.4, (E, n))
    pref = rng.uniform(-1.2, 1.2, (E, n, 2))
    vmax = rng.uniform(0.5, 1.5, (E, n))
    return pos, vel, rad, pref, vmax


@pytest.mark.parametrize("spread,n", [(4.0, 6), (1.5, 6), (0.7, 5), (0.35, 4), (3.0, 2), (2.0, 11)])
def test_vectorised_orca_equals_scalar_restatement(spread, n):
    """Sparse crowds (feasible programs), dense ones (most programs need linearProgram3) and overlapping agents
    (collision branch of the half-plane construction): same velocities as the one-agent-at-a-time restatement."""
    rng = np.random.default_rng(int(spread * 100) + n)
    E = 40
    pos, vel, rad, pref, vmax = _random_crowd(rng, E, n, spread, 1.0)
    got = EP.orca_velocities(pos, vel, rad, pref, vmax, time_horizon=2.0, time_step=0.25)
    n_lp3 = 0
    for e in range(E):
        for i in range(n):
            others = [((pos[e, j, 0], pos[e, j, 1]), (vel[e, j, 0], vel[e, j, 1]), rad[e, j]) for j in range(n) if j != i]
            lines = OO.orca_lines(tuple(pos[e, i]), tuple(vel[e, i]), rad[e, i], others, 2.0, 0.25)
            fail, _ = OO.linear_program2(lines, vmax[e, i], tuple(pref[e, i]), False)
            n_lp3 += fail < len(lines)
            want = OO.new_velocity(tuple(pos[e, i]), tuple(vel[e, i]), rad[e, i], tuple(pref[e, i]), vmax[e, i], others,
                                   2.0, 0.25)
            np.testing.assert_allclose(got[e, i], want, rtol=0, atol=1e-12, err_msg=f"episode {e} agent {i}")
    if spread <= 0.7:
        assert n_lp3 > 20          # the infeasible branch really ran


def test_feasible_programs_are_optimal_against_brute_force():
    """Where the half-planes leave a non-empty region, the result must be the point of (disc intersect half-planes)
    closest to the preferred velocity: checked against a dense polar grid of the disc."""
    rng = np.random.default_rng(5)
    E, n = 30, 5
    pos, vel, rad, pref, vmax = _random_crowd(rng, E, n, 3.0, 0.8)
    got = EP.orca_velocities(pos, vel, rad, pref, vmax)
    r = np.linspace(0.0, 1.0, 400)[:, None]
    th = np.linspace(0.0, 2 * np.pi, 1440, endpoint=False)[None, :]
    checked = 0
    for e in range(E):
        for i in range(n):
            others = [((pos[e, j, 0], pos[e, j, 1]), (vel[e, j, 0], vel[e, j, 1]), rad[e, j]) for j in range(n) if j != i]
            lines = OO.orca_lines(tuple(pos[e, i]), tuple(vel[e, i]), rad[e, i], others, 2.0, 0.25)
            fail, _ = OO.linear_program2(lines, vmax[e, i], tuple(pref[e, i]), False)
            if fail < len(lines):
                continue
            gx, gy = vmax[e, i] * r * np.cos(th), vmax[e, i] * r * np.sin(th)
            ok = np.ones_like(gx, dtype=bool)
            for (p, d) in lines:            # permitted side: det(direction, point - v) <= 0
                ok &= d[0] * (p[1] - gy) - d[1] * (p[0] - gx) <= 1e-9
                assert d[0] * (p[1] - got[e, i, 1]) - d[1] * (p[0] - got[e, i, 0]) <= 1e-9
            assert np.hypot(*got[e, i]) <= vmax[e, i] + 1e-9
            if ok.any():
                best = np.hypot(gx - pref[e, i, 0], gy - pref[e, i, 1])[ok].min()
                mine = np.hypot(got[e, i, 0] - pref[e, i, 0], got[e, i, 1] - pref[e, i, 1])
                assert mine <= best + 1e-9 and mine >= best - 0.02        # grid resolution
                checked += 1
    assert checked > 100


def test_circle_crossing_crowds_do_not_collide_and_cross():
    cfg = EP.CrowdConfig()
    E, N, steps = 64, 5, 60
    sim = EP.simulate_circle_crossing(E, N, steps, seed=3, cfg=cfg)
    hx, rx = sim["human_xy"], sim["robot_xy"]
    assert hx.shape == (E, steps + 1, N, 2) and rx.shape == (E, steps + 1, 2) and np.isfinite(hx).all()
    allp = np.concatenate([rx[:, :, None], hx], axis=2)                            # [E, F, n, 2]
    rad = sim["radius"]
    d = np.linalg.norm(allp[:, :, :, None] - allp[:, :, None, :], axis=-1)
    need = rad[:, None, :, None] + rad[:, None, None, :]
    iu = np.triu_indices(N + 1, 1)
    assert (d[:, :, iu[0], iu[1]] >= need[:, :, iu[0], iu[1]] - 1e-6).all()        # reciprocal avoidance: no contact
    speed = np.linalg.norm(np.diff(allp, axis=1), axis=-1) / cfg.time_step
    assert (speed <= sim["v_pref"][:, None, :] + 1e-9).all()
    # everybody makes it (nearly) across: final distance to the goal is a small fraction of the start distance
    d0 = np.linalg.norm(allp[:, 0] - sim["goal"], axis=-1)
    d1 = np.linalg.norm(allp[:, -1] - sim["goal"], axis=-1)
    assert np.median(d1 / d0) < 0.05 and (d1 < d0).all()
    # placement rule (crowd_sim_plus.py:472-477) and determinism
    start = allp[:, 0]
    ds = np.linalg.norm(start[:, :, None] - start[:, None, :], axis=-1)[:, iu[0], iu[1]]
    assert (ds >= (cfg.human_radius + rad[:, iu[0]] + cfg.discomfort_dist) - 1e-9).all()
    np.testing.assert_array_equal(EP.simulate_circle_crossing(E, N, steps, seed=3, cfg=cfg)["human_xy"], hx)
    assert not np.array_equal(EP.simulate_circle_crossing(E, N, steps, seed=4, cfg=cfg)["human_xy"], hx)


def test_histories_feed_the_batched_scene_builder():
    """The generator's output is what the predictor's host side consumes: natural clusters of varying size."""
    sim = EP.simulate_circle_crossing(96, 5, 24, seed=1)
    sizes = set()
    for frame in (8, 16, 24):
        hum, rob = EP.history_windows(sim, frame)
        b = SC.build_scenes_batched(hum, rob, 0.25, horizon=12)
        assert b["x_st"].shape == (96, 5, 6, 6) and np.isfinite(b["x_st"]).all() and np.isfinite(b["nbr_sum"]).all()
        sizes |= set(b["in_cluster"].sum(axis=1).tolist())
        e = 7
        sb = SC.build_scene(hum[e], rob[e], 0.25, 12)
        rows = np.nonzero(b["in_cluster"][e])[0]
        np.testing.assert_array_equal(b["x_st"][e, rows], sb.x_st)
    assert len(sizes) >= 3 and min(sizes) >= 1
    with pytest.raises(ValueError):
        EP.history_windows(sim, 3)


# ------------------------------------------------------------------------------------------------ reference pins
import glob
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "episodes_placement_*.npz"))))
def test_circle_crossing_placement_matches_the_reference_lines(case):
    """crowd_sim_plus.py:454-481 executed by tests/golden/make_golden_episodes.py on numpy's default_rng(seed): the same
    positions, goals and preferred speeds bit for bit, and the generator left in the same state (same number of draws)."""
    z = np.load(os.path.join(GOLDEN, case))
    cfg = EP.CrowdConfig(circle_radius=float(z["circle_radius"]), randomize_attributes=bool(z["randomize"]),
                         human_radius=float(z["human_radius"]), human_v_pref=float(z["human_v_pref"]),
                         robot_radius=float(z["robot_radius"]), discomfort_dist=float(z["discomfort_dist"]))
    rng = np.random.default_rng(int(z["seed"]))
    pos, goal, vp = EP.place_circle_crossing_humans(int(z["n_humans"]), rng, cfg)
    np.testing.assert_array_equal(pos, z["pos"])
    np.testing.assert_array_equal(goal, z["goal"])
    np.testing.assert_array_equal(vp, z["v_pref"])
    np.testing.assert_array_equal(rng.random(4), z["rng_next"])


@pytest.mark.parametrize("case", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "episodes_square_placement_*.npz"))))
def test_square_crossing_placement_matches_the_reference_lines(case):
    """crowd_sim_plus.py:490-520 executed by tests/golden/make_golden_episodes.py on numpy's default_rng(seed): the same starts
    (rejected against earlier STARTS only), goals (against earlier GOALS only) and preferred speeds bit for bit, and the
    generator left in the same state."""
    z = np.load(os.path.join(GOLDEN, case))
    cfg = EP.CrowdConfig(circle_radius=float(z["circle_radius"]), square_width=float(z["square_width"]),
                         randomize_attributes=bool(z["randomize"]), human_radius=float(z["human_radius"]),
                         human_v_pref=float(z["human_v_pref"]), robot_radius=float(z["robot_radius"]),
                         discomfort_dist=float(z["discomfort_dist"]))
    rng = np.random.default_rng(int(z["seed"]))
    pos, goal, vp = EP.place_square_crossing_humans(int(z["n_humans"]), rng, cfg)
    np.testing.assert_array_equal(pos, z["pos"])
    np.testing.assert_array_equal(goal, z["goal"])
    np.testing.assert_array_equal(vp, z["v_pref"])
    np.testing.assert_array_equal(rng.random(4), z["rng_next"])
    # every human crosses the y axis, inside the square
    assert np.all(pos[:, 0] * goal[:, 0] <= 0) and np.all(np.abs(np.concatenate([pos, goal])) <= cfg.square_width / 2)


def test_square_crossing_episodes_run_and_reach_their_goals():
    """The square-crossing rule through the batched simulator: deterministic per (seed, episode), no pair of agents ever closer
    than the sum of their radii, and most humans arrive (ORCA; the rule of crowd_sim_plus.py:436-439)."""
    cfg = EP.CrowdConfig(square_width=8.0, circle_radius=4.0)
    a = EP.simulate_crossing(6, 5, 60, seed=3, cfg=cfg, rule="square_crossing")
    b = EP.simulate_crossing(3, 5, 60, seed=3, cfg=cfg, rule="square_crossing")
    np.testing.assert_array_equal(a["human_xy"][:3], b["human_xy"])
    xy = np.concatenate([a["robot_xy"][:, :, None], a["human_xy"]], axis=2)            # [E, T, n, 2]
    d = np.linalg.norm(xy[:, :, :, None] - xy[:, :, None, :], axis=-1)
    rr = a["radius"][:, None, :, None] + a["radius"][:, None, None, :]
    iu = np.triu_indices(xy.shape[2], 1)
    assert np.all((d - rr)[:, :, iu[0], iu[1]] > -1e-6)
    left = np.linalg.norm(a["human_xy"][:, -1] - a["goal"][:, 1:], axis=-1)
    assert np.mean(left < 0.5) > 0.7
    with pytest.raises(ValueError):
        EP.crossing_starts(1, 2, 0, cfg, "hallway")


@pytest.mark.parametrize("case", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "episodes_orca_calls_*.npz"))))
def test_orca_call_parameters_match_what_the_reference_hands_to_rvo2(case):
    """orca.py:56-67, 93-129 executed against a recording rvo2 stand-in: simulator parameters, per-agent parameters (inflated
    radius, speed limit), positions / velocities in the reference's agent order and the ego's preferred velocity.  (What
    rvo2 then computes from them stays unpinned: it cannot run here.)"""
    z = np.load(os.path.join(GOLDEN, case))
    ego, oth = z["ego"], z["others"]
    n = 1 + len(oth)
    pos = np.concatenate([ego[None, 0:2], oth[:, 0:2]])[None]
    vel = np.concatenate([ego[None, 2:4], oth[:, 2:4]])[None]
    goal = np.concatenate([ego[None, 4:6], np.zeros((n - 1, 2))])[None]
    rad = np.concatenate([ego[6:7], oth[:, 4]])[None]
    vp = np.concatenate([ego[7:8], oth[:, 5]])[None]
    cfg = EP.CrowdConfig(time_step=float(z["time_step"]))
    par = EP.orca_call_parameters(cfg, pos, vel, goal, rad, vp)
    sim = z["simulator"]        # time_step, neighbor_dist, max_neighbors, time_horizon, time_horizon_obst, radius, max_speed
    assert [par["time_step"], par["neighbor_dist"], par["max_neighbors"], par["time_horizon"], par["time_horizon_obst"],
            par["default_radius"], par["default_max_speed"]] == sim.tolist()
    ap = z["agent_params"]      # per agent: neighbor_dist, max_neighbors, time_horizon, time_horizon_obst, radius, max_speed
    assert (ap[:, 0] == par["neighbor_dist"]).all() and (ap[:, 1] == par["max_neighbors"]).all()
    assert (ap[:, 2] == par["time_horizon"]).all() and (ap[:, 3] == par["time_horizon_obst"]).all()
    np.testing.assert_array_equal(par["radius"][0], ap[:, 4])                    # radius + 0.01 + safety space, every agent
    assert par["max_speed"][0, 0] == ap[0, 5]                                     # the ego moves at up to ITS v_pref
    assert (ap[1:, 5] == cfg.orca_default_max_speed).all()                        # (the others' limit is never read: their
    np.testing.assert_array_equal(z["agent_pos"], pos[0])                        #  velocities are not the call's output)
    np.testing.assert_array_equal(z["agent_vel"], vel[0])
    np.testing.assert_array_equal(par["pref"][0, 0], z["pref"][0])               # the ego's preferred velocity
    assert (z["pref"][1:] == 0).all()                                             # everybody else: (0, 0), orca.py:123-125
    if "near_goal" in case:
        assert np.linalg.norm(z["pref"][0]) < 1.0                                 # not normalised inside 1 m


def test_only_the_nearest_neighbours_within_range_constrain_an_agent():
    """RVO2 keeps the max_neighbors nearest agents closer than neighbor_dist (the reference passes 10 and 10 m): a far
    agent, or an eleventh one, must not change the result."""
    rng = np.random.default_rng(17)
    pos, vel, rad, pref, vmax = _random_crowd(rng, 6, 4, 2.0, 0.8)
    base = EP.orca_velocities(pos, vel, rad, pref, vmax)
    far = np.concatenate([pos, pos[:, :1] + np.array([30.0, 0.0])], axis=1)         # a fifth agent 30 m away
    ext = lambda a, v: np.concatenate([a, np.full_like(a[:, :1], v)], axis=1)
    got = EP.orca_velocities(far, ext(vel, 0.3), ext(rad, 0.3), ext(pref, 0.0), ext(vmax, 1.0))
    np.testing.assert_allclose(got[:, :4], base, rtol=0, atol=1e-12)
    pos13, vel13, rad13, pref13, vmax13 = _random_crowd(rng, 3, 13, 3.0, 0.8)
    full = EP.orca_velocities(pos13, vel13, rad13, pref13, vmax13, max_neighbors=100)
    lim = EP.orca_velocities(pos13, vel13, rad13, pref13, vmax13, max_neighbors=10)
    for e in range(3):
        for i in range(13):
            d = np.linalg.norm(pos13[e] - pos13[e, i], axis=1)
            keep = np.argsort(d, kind="stable")[:11]                                 # itself + its 10 nearest
            want = EP.orca_velocities(pos13[e:e + 1, keep], vel13[e:e + 1, keep], rad13[e:e + 1, keep],
                                      pref13[e:e + 1, keep], vmax13[e:e + 1, keep], max_neighbors=100)[0, 0]
            np.testing.assert_allclose(lim[e, i], want, rtol=0, atol=1e-12)
    assert not np.allclose(full, lim)
