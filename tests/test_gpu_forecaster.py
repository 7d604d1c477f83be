"""The drop-in class surface against the reference's captured end-to-end results
(tests/golden/wrapper_*.npz: update_state_hists x n -> predict_ret_best())."""
import glob
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from safe_interactive_crowdnav_amd.forecaster import HumanTrajectoryForecasterSim, write_configs
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PROD_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "safe-interactive-crowdnav_amd", "csrc", "libjmid_hip.so")
CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "wrapper_*.npz")))


class State:
    def __init__(self, p):
        self.position = (float(p[0]), float(p[1]))


@pytest.mark.parametrize("flavour", ["diag", "prod"])
@pytest.mark.parametrize("case", CASES)
def test_predict_ret_best_matches_reference(case, flavour, tmp_path):
    z = np.load(os.path.join(GOLDEN, case))
    N, K, k_ret, H = int(z["N"]), int(z["K"]), int(z["k_ret"]), int(z["H"])
    env, ypath = write_configs(str(tmp_path), joint=bool(z["joint"]), ctx_dim=int(z["ctx_dim"]), N=N, K=K,
                               k_ret=k_ret, H=H, step=int(z["step"]), time_step=float(z["time_step"]))
    w = JMIDWeights.from_seed(NetDims(ctx_dim=int(z["ctx_dim"])), int(z["wseed"]))
    assert w.checksum() == str(z["wsum"])
    # the captures are CPU-reference runs; flavour "prod" = the production libjmid_hip.so next to the session's diagnostics build,
    # constructed as the drop-in user constructs it (the class default: "f16mx" + self_check); "diag" = the fp32-class mode by name
    f = HumanTrajectoryForecasterSim(env, ypath, weights=w, rng_compat="cpu", lib_path=PROD_LIB if flavour == "prod" else None,
                                     precision=None if flavour == "prod" else "f16x3")
    assert f.precision == ("f16mx" if flavour == "prod" else "f16x3") and f.self_check == (flavour == "prod")
    assert f.engine._lib.has_diagnostics == (flavour == "diag")
    assert f.num_hist_frames == int(z["past"])
    for r, h, t in zip(z["robot_xy"], z["human_xy"], z["stamps"]):
        f.update_state_hists(State(r), [State(p) for p in h], float(t))
    torch.manual_seed(int(z["dseed"]))
    forecasts, logw = f.predict_ret_best()
    assert forecasts.shape == (N, k_ret, H + 1, 2) and forecasts.dtype == np.float64
    assert logw.shape == (N, k_ret) and logw.dtype == np.float64
    if k_ret >= K or "tight" in case:
        # all samples returned - or wrapper_jmid_topk_tight: a 100 Hz environment whose samples lie within the KDE bandwidths of
        # each other, so the joint-KDE ranking is not a tie and the reference's top-k CHOICE AND ORDER must be reproduced
        if "tight" in case:
            assert np.diff(np.sort(z["logw"][0])).min() > 1e-3 and np.ptp(z["logw"][0]) > 0.5
        ade = np.linalg.norm(forecasts - z["forecasts"], axis=-1).mean()
        print(f"{case}: mean ADE(forecasts) vs reference = {ade:.3e}")
        assert ade <= 1e-4
        np.testing.assert_allclose(logw, z["logw"], rtol=0, atol=1e-3)
    else:
        # top-k selection (mid_sim_wrapper.py:487-492).  With random-init weights the K samples are so spread
        # out that the joint KDE (bandwidth 0.01-0.1 m) gives every sample the same likelihood: the reference's
        # choice is then an arbitrary tie-break of torch.argsort.  The KDE itself is pinned bit-exactly on
        # non-degenerate inputs in tests/test_host_logic.py; here the selection is checked as a set: every
        # returned joint sample must be one of the reference's K samples, all distinct, with the same weights.
        ids = np.sort(z["node_ids"])
        order = np.argsort(z["node_ids"])
        pos_all = np.cumsum(z["vel"].astype(np.float64), axis=2) * float(z["time_step"]) \
            + z["x_t"][:, -1, 0:2].astype(np.float64)[None, :, None, :]
        pos_all = pos_all[:, order]                                        # [K, A, H, 2]
        got = forecasts[ids][:, :, 1:, :].transpose(1, 0, 2, 3)            # [k, A, H, 2]
        dist = np.linalg.norm(got[:, None] - pos_all[None], axis=-1).mean(axis=(2, 3))   # [k, K]
        match = dist.argmin(axis=1)
        assert len(set(match.tolist())) == k_ret
        assert dist.min(axis=1).max() <= 1e-4
        np.testing.assert_allclose(np.sort(logw, axis=1), np.sort(z["logw"], axis=1), rtol=0, atol=1e-3)
        np.testing.assert_allclose(forecasts[:, :, 0, :], z["forecasts"][:, :, 0, :], rtol=0, atol=0)
    # RNG contract: the global CPU generator is left where the (CPU-only) reference leaves it
    torch.manual_seed(int(z["dseed"]))
    A = len(z["node_ids"])
    x_T = torch.randn([K * A, H, 2])
    for _ in range(int(z["step"])):
        torch.randn_like(x_T)
    expect_next = torch.randn(4)
    torch.manual_seed(int(z["dseed"]))
    f.predict_ret_best()
    assert torch.equal(torch.randn(4), expect_next)


def test_rng_compat_cuda_leaves_the_cpu_generator_after_x_T(tmp_path):
    """On a GPU host the reference draws its per-step z on the device x_T was moved to (MID/models/diffusion.py:499-509,
    MID/mid.py:91): the CPU generator advances by x_T only.  rng_compat="cuda" (and "auto" here) reproduces that; the
    forecasts depend on x_T alone, so they equal the rng_compat="cpu" ones."""
    z = np.load(os.path.join(GOLDEN, "wrapper_jmid_w32_spread50.npz"))
    N, K, H = int(z["N"]), int(z["K"]), int(z["H"])
    env, ypath = write_configs(str(tmp_path), joint=True, ctx_dim=32, N=N, K=K, k_ret=int(z["k_ret"]), H=H, step=2)
    w = JMIDWeights.from_seed(NetDims(ctx_dim=32), int(z["wseed"]))
    outs = {}
    for mode in ("cpu", "cuda", "auto"):
        f = HumanTrajectoryForecasterSim(env, ypath, weights=w, rng_compat=mode)
        assert f.rng_compat == ("cuda" if mode == "auto" else mode)
        for r, h, t in zip(z["robot_xy"], z["human_xy"], z["stamps"]):
            f.update_state_hists(State(r), [State(p) for p in h], float(t))
        torch.manual_seed(7)
        outs[mode] = f.predict_ret_best()[0]
        after = torch.randn(4)
        torch.manual_seed(7)
        x_T = torch.randn([K * len(z["node_ids"]), H, 2])
        if mode == "cpu":
            for _ in range(2):
                torch.randn_like(x_T)
        assert torch.equal(torch.randn(4), after), mode
    np.testing.assert_array_equal(outs["cpu"], outs["cuda"])
    np.testing.assert_array_equal(outs["cpu"], outs["auto"])
    with pytest.raises(ValueError):
        HumanTrajectoryForecasterSim(env, ypath, weights=w, rng_compat="numpy")


def test_device_generator_is_advanced_exactly_as_by_the_reference_draws():
    """rng_compat="cuda": the reference's per-step z (MID/models/diffusion.py:509, unused by DDIM) are not launched; the device
    generator's Philox offset is moved by what they would have consumed.  State and next draw must equal the real thing - on the
    call that learns a shape's step (one real draw) and on the ones after it, for shapes of every launch geometry."""
    from safe_interactive_crowdnav_amd import forecaster as F
    for shape, n in (((100, 12, 2), 49), ((300, 8, 2), 1), ((7, 3, 2), 5), ((4096, 24, 2), 3), ((1, 1, 2), 2)):
        F._PHILOX_STEP.pop((0, shape), None)
        for attempt in ("learns", "knows"):
            torch.cuda.manual_seed(1234)
            z = torch.empty(shape, device="cuda:0")
            for _ in range(n):
                torch.randn_like(z)
            state = torch.cuda.get_rng_state(0)
            nxt = torch.randn(5, device="cuda:0")
            torch.cuda.manual_seed(1234)
            F.advance_cuda_generator(0, shape, n)
            assert torch.equal(torch.cuda.get_rng_state(0), state), (shape, attempt)
            assert torch.equal(torch.randn(5, device="cuda:0"), nxt), (shape, attempt)
        assert (0, shape) in F._PHILOX_STEP


def test_returned_arrays_are_fresh(tmp_path):
    z = np.load(os.path.join(GOLDEN, "wrapper_jmid_w32_spread50.npz"))
    env, ypath = write_configs(str(tmp_path), joint=True, ctx_dim=32, N=int(z["N"]), K=int(z["K"]),
                               k_ret=int(z["k_ret"]), H=int(z["H"]), step=2)
    f = HumanTrajectoryForecasterSim(env, ypath, weights=JMIDWeights.from_seed(NetDims(ctx_dim=32), int(z["wseed"])))
    for r, h, t in zip(z["robot_xy"], z["human_xy"], z["stamps"]):
        f.update_state_hists(State(r), [State(p) for p in h], float(t))
    a, _ = f.predict_ret_best()
    keep = a.copy()
    b, _ = f.predict_ret_best()
    assert a is not b and np.array_equal(a, keep)    # caller-owned results (sicnav_acados.py:1652 stores them)


@pytest.mark.expects_erange
def test_forecaster_falls_back_to_fp32_when_fp16_range_is_exceeded(tmp_path):
    """Histories a kilometre-scale apart push the standardized inputs and hence activations... not the fp16 range by
    themselves; force it with weights whose first layer is scaled up, and check that predict_ret_best() still answers
    (exact-fp32 retry) and equals a forecaster that was asked for fp32 from the start."""
    z = np.load(os.path.join(GOLDEN, "wrapper_jmid_w32_spread50.npz"))
    w = JMIDWeights.from_seed(NetDims(ctx_dim=32), int(z["wseed"]))
    t = dict(w.tensors)
    t["transformer_encoder.layers.0.linear1.weight"] = t["transformer_encoder.layers.0.linear1.weight"] * 4e4
    t["transformer_encoder.layers.0.linear2.weight"] = t["transformer_encoder.layers.0.linear2.weight"] / 4e4
    wbig = JMIDWeights(w.dims, t)
    outs = []
    for prec in ("f16x3", "f16x2", "f16mx", "f32"):
        env, ypath = write_configs(str(tmp_path / prec), joint=True, ctx_dim=32, N=int(z["N"]), K=int(z["K"]),
                                   k_ret=int(z["k_ret"]), H=int(z["H"]), step=2)
        f = HumanTrajectoryForecasterSim(env, ypath, weights=wbig, precision=prec)
        for r, h, tt in zip(z["robot_xy"], z["human_xy"], z["stamps"]):
            f.update_state_hists(State(r), [State(p) for p in h], float(tt))
        torch.manual_seed(1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            outs.append(f.predict_ret_best()[0])
        assert f.erange_fallbacks == (0 if prec == "f32" else 1)      # the slow path is counted (and warned about once)
    assert np.isfinite(outs[0]).all()
    np.testing.assert_array_equal(outs[0], outs[2])
    np.testing.assert_array_equal(outs[1], outs[2])


@pytest.mark.parametrize("ctx_dim,joint,k_ret", [(32, True, 16), (32, False, 16), (256, True, 16), (32, True, 5)])
def test_predict_batch_with_natural_clusters_matches_per_episode_forecaster(ctx_dim, joint, k_ret, tmp_path):
    """SURVEY 8f row f2 on the device path: a multi-episode batch with the reference's own clustering (the number of
    in-cluster pedestrians differs from episode to episode) through build_scenes_batched -> engine, against one
    HumanTrajectoryForecasterSim per episode fed the same histories and the same seed."""
    from safe_interactive_crowdnav_amd.forecaster import predict_batch
    E, F, N, K, H, dt = 24, 6, 6, 16, 8, 0.25
    rng = np.random.default_rng(17)
    pos0 = rng.uniform(-5.0, 5.0, (E, N, 2))
    vel = rng.uniform(-1.0, 1.0, (E, N, 2))
    t = np.arange(F) * dt
    hum = pos0[:, None] + vel[:, None] * t[None, :, None, None] + 0.01 * rng.standard_normal((E, F, N, 2))
    rob = np.array([0.0, -3.0])[None, None] + 0.02 * rng.standard_normal((E, F, 2))
    w = JMIDWeights.from_seed(NetDims(ctx_dim=ctx_dim), 5)
    env, ypath = write_configs(str(tmp_path), joint=joint, ctx_dim=ctx_dim, N=N, K=K, k_ret=k_ret, H=H, step=2)
    ref_f, ref_w, ref_in = [], [], []
    for e in range(E):
        f = HumanTrajectoryForecasterSim(env, ypath, weights=w)
        for i in range(F):
            f.update_state_hists(State(rob[e, i]), [State(p) for p in hum[e, i]], float(t[i]))
        torch.manual_seed(1000 + e)
        a, b = f.predict_ret_best()
        ref_f.append(a)
        ref_w.append(b)
    eng = f.engine
    fc, lw, inc = predict_batch(eng, hum, rob, [1000 + e for e in range(E)], num_samples=K, num_ret_samples=k_ret,
                                horizon=H, time_step=dt, precision=f.precision)
    sizes = inc.sum(axis=1)
    assert len(np.unique(sizes)) >= 3 and sizes.min() >= 1 and sizes.max() <= N      # a genuinely ragged batch
    assert fc.shape == (E, N, k_ret, H + 1, 2) and fc.dtype == np.float64 and lw.shape == (E, N, k_ret)
    ref_f, ref_w = np.stack(ref_f), np.stack(ref_w)
    if ctx_dim == 32:
        # head_dim 16: one episode alone and the same episode inside a group run the very same arithmetic
        np.testing.assert_array_equal(fc, ref_f)
        np.testing.assert_array_equal(lw, ref_w)
    else:
        # head_dim 128: the split-KV factor of attention depends on the episode count of the call (include/jmid_hip.h)
        assert np.linalg.norm(fc - ref_f, axis=-1).mean() <= 1e-6
        np.testing.assert_allclose(lw, ref_w, rtol=0, atol=1e-9)
    out = ~inc
    np.testing.assert_array_equal(fc[out], ref_f[out])          # constant-velocity rows: host arithmetic, bit-exact


def test_predictor_output_feeds_the_mpc_parameter_layout(tmp_path):
    """SURVEY 8f row f1 end to end on the device path: predict_ret_best() of the drop-in class -> mpc_glue -> the
    per-stage Acados parameter vectors, against the reference capture's forecasts for the same scene (the wrapper
    golden holds what the reference's predictor returned; the arithmetic after it is pinned bit-exactly on the host by
    tests/test_host_logic.py::test_mpc_glue_matches_reference_capture)."""
    import einops
    from safe_interactive_crowdnav_amd.mpc_glue import mpc_forecast_inputs, stage_parameter_blocks
    z = np.load(os.path.join(GOLDEN, "wrapper_jmid_together.npz"))
    N, K, k_ret, H = int(z["N"]), int(z["K"]), int(z["k_ret"]), int(z["H"])
    env, ypath = write_configs(str(tmp_path), joint=True, ctx_dim=int(z["ctx_dim"]), N=N, K=K, k_ret=k_ret, H=H,
                               step=int(z["step"]))
    f = HumanTrajectoryForecasterSim(env, ypath, weights=JMIDWeights.from_seed(NetDims(ctx_dim=int(z["ctx_dim"])),
                                                                                 int(z["wseed"])), rng_compat="cpu")
    for r, h, t in zip(z["robot_xy"], z["human_xy"], z["stamps"]):
        f.update_state_hists(State(r), [State(p) for p in h], float(t))
    torch.manual_seed(int(z["dseed"]))
    top, w = f.predict_ret_best()
    horiz = H - 1
    rng = np.random.default_rng(0)
    nx, nu = 4 + 4 * N, 2
    gs, ga = rng.standard_normal((nx, horiz + 1)), rng.standard_normal((nu, horiz))
    Q, R, TQ = rng.uniform(0.1, 1, nx), rng.uniform(0.1, 1, nu), rng.uniform(0.1, 1, nx)
    ours = mpc_forecast_inputs(top, w, horiz, f.time_step, joint=True)
    ref = mpc_forecast_inputs(z["forecasts"], z["logw"], horiz, f.time_step, joint=True)
    p = stage_parameter_blocks(ours.samples_by_stage, gs, ga, Q, R, TQ, horiz)
    p_ref = stage_parameter_blocks(ref.samples_by_stage, gs, ga, Q, R, TQ, horiz)
    assert p.shape == (horiz + 1, nx + nu + 2 * nx + nu + 4 * N * k_ret)
    np.testing.assert_array_equal(ours.samples_by_stage,
                                  einops.rearrange(top[:, :, 1:, :], "h s t d -> t (h s) d")[: horiz + 1])
    assert np.abs(p - p_ref).max() <= 5e-4 and np.abs(p - p_ref).mean() <= 1e-4     # forecasts within the ADE gate
    np.testing.assert_allclose(ours.goal_xy, ref.goal_xy, atol=1e-4)
    np.testing.assert_allclose(ours.v_pref, ref.v_pref, atol=2e-3)


def test_predict_batch_on_generated_orca_crowds(tmp_path):
    """SURVEY 8f rows f3 + f2 together on the device path: histories from the batched ORCA circle-crossing generator
    (episodes.py), the reference's clustering, through the engine - against one forecaster per episode."""
    from safe_interactive_crowdnav_amd.episodes import history_windows, simulate_circle_crossing
    from safe_interactive_crowdnav_amd.forecaster import predict_batch
    E, N, K, H, dt = 16, 5, 12, 8, 0.25
    sim = simulate_circle_crossing(E, N, 14, seed=9)
    hum, rob = history_windows(sim, 14)
    w = JMIDWeights.from_seed(NetDims(ctx_dim=32), 5)
    env, ypath = write_configs(str(tmp_path), joint=True, ctx_dim=32, N=N, K=K, k_ret=K, H=H, step=2)
    ref = []
    for e in range(E):
        f = HumanTrajectoryForecasterSim(env, ypath, weights=w)
        for i in range(hum.shape[1]):
            f.update_state_hists(State(rob[e, i]), [State(p) for p in hum[e, i]], float(sim["stamps"][14 - 5 + i]))
        torch.manual_seed(77 + e)
        ref.append(f.predict_ret_best()[0])
    fc, lw, inc = predict_batch(f.engine, hum, rob, [77 + e for e in range(E)], num_samples=K, num_ret_samples=K,
                                horizon=H, time_step=dt)
    assert len(set(inc.sum(axis=1).tolist())) >= 2
    np.testing.assert_array_equal(fc, np.stack(ref))


def test_predict_batch_on_generated_hallway_crowds(tmp_path):
    """The reference's SHIPPED scenario end to end on the device path: histories from the batched hallway generator
    (crowd_env.simulate_hallway: walls, orca_plus humans, ten steps of head start; env.config [sim]), the reference's clustering
    (in a 1.75 m corridor nearly everybody is within 3 m of the robot), through the engine at the shipped shape
    (N = 3, K = 100 -> 15 kept, H = 8, 2 denoise steps) - against one forecaster per episode."""
    from safe_interactive_crowdnav_amd.crowd_env import simulate_hallway
    from safe_interactive_crowdnav_amd.episodes import history_windows
    from safe_interactive_crowdnav_amd.forecaster import predict_batch
    E, N, K, k, H, dt = 12, 3, 100, 15, 8, 0.25
    sim = simulate_hallway(E, N, 10, seed=21)
    hum, rob = history_windows(sim, 10)
    w = JMIDWeights.from_seed(NetDims(ctx_dim=32), 6)
    env, ypath = write_configs(str(tmp_path), joint=True, ctx_dim=32, N=N, K=K, k_ret=k, H=H, step=2)
    ref, refw = [], []
    for e in range(E):
        f = HumanTrajectoryForecasterSim(env, ypath, weights=w)
        for i in range(hum.shape[1]):
            f.update_state_hists(State(rob[e, i]), [State(p) for p in hum[e, i]], float(sim["stamps"][10 - 5 + i]))
        torch.manual_seed(31 + e)
        fc1, lw1 = f.predict_ret_best()
        ref.append(fc1)
        refw.append(lw1)
    fc, lw, inc = predict_batch(f.engine, hum, rob, [31 + e for e in range(E)], num_samples=K, num_ret_samples=k, horizon=H, time_step=dt)
    assert fc.shape == (E, N, k, H + 1, 2) and inc.any()
    np.testing.assert_array_equal(fc, np.stack(ref))
    np.testing.assert_array_equal(lw, np.stack(refw))
