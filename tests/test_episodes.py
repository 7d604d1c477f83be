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
