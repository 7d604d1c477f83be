"""Host batch builder (scene.py) against the reference's own batch tensors captured in tests/golden/wrapper_*.npz."""
import glob
import os

import numpy as np
import pytest

from safe_interactive_crowdnav_amd import scene as SC

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "wrapper_*.npz")))


def replay_histories(z):
    """update_state_hists (mid_sim_wrapper.py:198-204): per-human FIFO capped at past_num_frames, robot unbounded."""
    N, past = int(z["N"]), int(z["past"])
    prev = [[] for _ in range(N)]
    rob = []
    for r, h, t in zip(z["robot_xy"], z["human_xy"], z["stamps"]):
        for i in range(N):
            prev[i].append([h[i, 0], h[i, 1], t])
            if len(prev[i]) > past:
                prev[i].pop(0)
        rob.append([r[0], r[1], t])
    return prev, rob


@pytest.mark.parametrize("case", CASES)
def test_batch_tensors_match_reference(case):
    z = np.load(os.path.join(GOLDEN, case))
    prev, rob = replay_histories(z)
    hum_xy, rob_xy, pose_now = SC.frame_table(prev, rob, float(z["time_step"]), int(z["past"]))
    sb = SC.build_scene(hum_xy, rob_xy, float(z["time_step"]), int(z["H"]), int(z["past"]))
    np.testing.assert_array_equal(sb.ids_in, z["node_ids"])
    np.testing.assert_allclose(sb.x, z["x_t"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(sb.x_st, z["x_st"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(sb.nbr_sum, z["nbr_sum"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(sb.edge_mask, z["edge_mask"], rtol=0, atol=1e-7)
    # current pose prepended to every forecast (mid_sim_wrapper.py:444-454)
    np.testing.assert_allclose(z["forecasts"][:, 0, 0, :], pose_now, rtol=0, atol=0)
    # constant-velocity rows for the pedestrians outside the chosen cluster (mid_sim_wrapper.py:413-429, 496-504)
    for i in sb.ids_out:
        np.testing.assert_allclose(z["forecasts"][i, 0, 1:, :], sb.cv_forecasts[int(i)], rtol=0, atol=1e-12)
    assert sorted(list(sb.ids_in) + list(sb.ids_out)) == list(range(int(z["N"])))


def test_resampling_matches_pandas_semantics():
    # stamps from the probe in tests/golden/make_golden.py (jitter case): keep-last per end-anchored bin,
    # empty bins linearly interpolated (mid_sim_wrapper.py:283-298)
    t = [0.0, 0.22, 0.522, 0.775, 1.017, 1.245, 1.48, 1.725, 2.0]
    prev = [[[float(i), 10.0 * i, ti] for i, ti in enumerate(t)]]
    rob = [[0.0, 0.0, ti] for ti in t]
    h, r, pose = SC.frame_table(prev, rob, 0.25, 99)
    np.testing.assert_allclose(h[:, 0, 0], [0.0, 1.0, 1.5, 2.0, 3.0, 5.0, 6.0, 7.0, 8.0])
    assert pose.tolist() == [[8.0, 80.0]]


def test_short_history_raises():
    prev = [[[0.0, 0.0, 0.25 * i] for i in range(3)]]
    rob = [[1.0, 1.0, 0.25 * i] for i in range(3)]
    h, r, _ = SC.frame_table(prev, rob, 0.25, 6)
    with pytest.raises(SC.HistoryTooShortError):
        SC.build_scene(h, r, 0.25, 12, 6)


@pytest.mark.parametrize("force", [True, False])
def test_batched_builder_matches_per_episode_builder(force):
    """the vectorised multi-episode builder is bit-identical to build_scene per episode
    (which is itself pinned to the reference captures above)."""
    rng = np.random.default_rng(7)
    E, F, N = 40, 6, 5
    pos0 = rng.uniform(-3.5, 3.5, (E, N, 2))
    vel = rng.uniform(-1.0, 1.0, (E, N, 2))
    t = np.arange(F) * 0.25
    hum = pos0[:, None] + vel[:, None] * t[None, :, None, None] + 0.01 * rng.standard_normal((E, F, N, 2))
    rob = rng.uniform(-3, 3, (E, 1, 2)) + np.array([0.0, 0.2])[None, None] * t[None, :, None]
    b = SC.build_scenes_batched(hum, rob, 0.25, force_all_in_cluster=force)
    n_checked = 0
    for e in range(E):
        sb = SC.build_scene(hum[e], rob[e], 0.25, 12, F, force_all_in_cluster=force)
        ids = sb.ids_in
        np.testing.assert_array_equal(np.nonzero(b["in_cluster"][e])[0], ids)
        assert bool(b["robot_in_cluster"][e]) == sb.robot_in_cluster
        for key in ("x", "x_st", "nbr_sum", "edge_mask", "p0"):
            np.testing.assert_array_equal(b[key][e][ids], getattr(sb, key), err_msg=f"episode {e} {key}")
        n_checked += len(ids)
    assert n_checked > E            # clusters are non-trivial


def test_synthetic_episodes_shapes_and_determinism():
    a = SC.synthetic_episodes(6, 5, seed=3)
    b = SC.synthetic_episodes(6, 5, seed=3)
    assert a["x_st"].shape == (6, 5, 6, 6) and a["nbr_sum"].shape == (6, 5, 2, 6, 6) and a["gt"].shape == (6, 5, 12, 2)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    # constant-velocity histories: ground truth continues the last frame
    np.testing.assert_allclose(a["gt"][:, :, 0] - a["p0"], (a["x"][:, :, -1, 2:4] * 0.25), atol=1e-5)
