"""Host-side logic that needs no GPU: KDE top-k, weight container round trip, DDIM step table."""
import glob
import os

import numpy as np
import pytest
import torch

from safe_interactive_crowdnav_amd.kde import most_likely_samples
from safe_interactive_crowdnav_amd.schedule import VarianceSchedule, ddim_steps
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims, all_shapes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "kde_*.npz"))))
def test_kde_topk_matches_reference(case):
    z = np.load(os.path.join(GOLDEN, case))
    top, lw = most_likely_samples(z["forecasts"], int(z["k_ret"]))
    np.testing.assert_array_equal(top, z["top"])
    np.testing.assert_allclose(lw, z["logw"], rtol=1e-5, atol=1e-5)


def test_schedule_matches_reference_buffers():
    z = np.load(os.path.join(GOLDEN, "schedule.npz"))
    s = VarianceSchedule.linear()
    np.testing.assert_array_equal(s.betas, z["betas"])
    np.testing.assert_array_equal(s.alpha_bars, z["alpha_bars"])
    np.testing.assert_array_equal(s.sigmas_inflex, z["sigmas_inflex"])
    np.testing.assert_array_equal(s.sigmas_flex, z["sigmas_flex"])


def test_ddpm_table_flexibility():
    """get_sigmas (diffusion.py:59-64): sigma = flex * sigmas_flex[t] + (1 - flex) * sigmas_inflex[t]."""
    from safe_interactive_crowdnav_amd.schedule import ddpm_steps
    s = VarianceSchedule.linear()
    t0, t1, th = ddpm_steps(s, 10, 0.0), ddpm_steps(s, 10, 1.0), ddpm_steps(s, 10, 0.5)
    assert [x.t for x in t0] == list(range(100, 0, -10)) and all(x.noise for x in t0)
    assert [x.sigma for x in t0] == [s.sigmas_inflex[x.t] for x in t0]
    assert [x.sigma for x in t1] == [s.sigmas_flex[x.t] for x in t1]
    assert all(min(a.sigma, b.sigma) <= h.sigma <= max(a.sigma, b.sigma) for a, b, h in zip(t0, t1, th))
    assert not ddpm_steps(s, 100)[-1].noise            # t == 1: z = 0 (diffusion.py:571)
    with pytest.raises(ValueError):
        ddpm_steps(s, 10, 1.5)


def test_ddim_table():
    s = VarianceSchedule.linear()
    tab = ddim_steps(s, 50)
    assert [x.t for x in tab] == list(range(100, 0, -2))
    assert tab[-1].n_x == np.float32(1.0) and tab[-1].n_e == np.float32(0.0)   # abar_0 = 1: last step returns x0
    assert [x.t for x in ddim_steps(s, 2)] == [100, 50]
    with pytest.raises(ValueError):
        ddim_steps(s, 30)   # stride 3 does not divide 100 (reference would wrap around)


def test_weight_container_roundtrip(tmp_path):
    dims = NetDims(ctx_dim=32)
    w = JMIDWeights.from_seed(dims, 3)
    assert list(w.tensors) == list(all_shapes(dims))
    assert w.checksum() == JMIDWeights.from_seed(dims, 3).checksum() != JMIDWeights.from_seed(dims, 4).checksum()
    p = str(tmp_path / "w.npz")
    w.save(p)
    w2 = JMIDWeights.load(p)
    assert w2.dims == dims and w2.checksum() == w.checksum()
    n_live = sum(int(np.prod(s)) for k, s in all_shapes(NetDims()).items() if "/" not in k)
    assert n_live == 6_940_432    # live parameter count of the net
    with pytest.raises(KeyError):
        JMIDWeights(dims, {})


def test_reference_state_import():
    """from_reference_state accepts the checkpoint's own key layout (vel_predictor.net.* + ModuleDict)."""
    dims = NetDims(ctx_dim=32)
    w = JMIDWeights.from_seed(dims, 1)
    net_state = {"vel_predictor.net." + k: v for k, v in w.net_state_dict().items()}
    net_state["vel_predictor.net.layer.linear1.weight"] = torch.zeros(1)   # the unused template copy is ignored
    mods = {}
    for name, sd in w.encoder_state_dicts().items():
        if "edge_influence" in name:
            m = torch.nn.Module()
            m.w1 = torch.nn.Linear(16, 16, bias=False)
            m.w2 = torch.nn.Linear(16, 16, bias=False)
            m.v = torch.nn.Linear(16, 1, bias=False)
        else:
            m = torch.nn.LSTM(sd["weight_ih_l0"].shape[1], 16, batch_first=True)
        m.load_state_dict(sd)
        mods[name] = m
    w2 = JMIDWeights.from_reference_state(dims, net_state, mods)
    assert w2.checksum() == w.checksum()


def test_mpc_glue_matches_caller_arithmetic():
    """same arithmetic as SICNavAcados.predict (sicnav_acados.py:1644-1667), written there with
    einops and per-human loops."""
    import einops
    from safe_interactive_crowdnav_amd.mpc_glue import mpc_forecast_inputs

    rng = np.random.default_rng(0)
    N, k, H, horiz, dt = 3, 15, 8, 5, 0.25
    top = rng.standard_normal((N, k, H + 1, 2))
    w = np.log(rng.dirichlet(np.ones(k)))[None].repeat(N, axis=0)
    out = mpc_forecast_inputs(top, w, horiz, dt, joint=True)
    forecasts = top[:, :, 1:, :]
    np.testing.assert_array_equal(out.samples_by_stage,
                                  einops.rearrange(forecasts, "h s t d -> t (h s) d")[: horiz + 1, :, :])
    np.testing.assert_array_equal(out.init_weights, w[0, :])
    for h in range(N):
        assert out.goal_xy[h, 0] == pytest.approx(np.mean(forecasts[h, :, 0, 0]))
        assert out.goal_xy[h, 1] == pytest.approx(np.mean(forecasts[h, :, 0, 1]))
        assert out.v_pref[h] == pytest.approx(np.max(np.linalg.norm(np.diff(forecasts[h], axis=1), axis=2) / dt))
    assert mpc_forecast_inputs(top, w, horiz, dt, joint=False).init_weights.shape == (N, k)


@pytest.mark.parametrize("case", ["mpc_glue_joint.npz", "mpc_glue_indep_outdoor.npz", "mpc_glue_exact_horizon.npz"])
def test_mpc_glue_matches_reference_capture(case, golden_dir):
    """SURVEY 8f row f1 pinned by captures of the reference's own lines (tests/golden/make_golden_mpc.py executes
    sicnav_acados.py:1639-1682 and :1388-1413 against recording stand-ins): forecast slicing, initial weights, the
    't (h s) d' stage layout, goal point / preferred speed / heading per human, and the per-stage Acados parameter
    vectors p_0 .. p_horiz, bit for bit."""
    import os
    from safe_interactive_crowdnav_amd.mpc_glue import human_headings, mpc_forecast_inputs, stage_parameter_blocks

    z = np.load(os.path.join(golden_dir, case))
    horiz, joint, outdoor = int(z["horiz"]), bool(z["joint"]), bool(z["outdoor"])
    out = mpc_forecast_inputs(z["top_k_forecasts"], z["top_k_weights"], horiz, float(z["time_step"]), joint=joint)
    np.testing.assert_array_equal(out.forecasts, z["forecasts"])
    np.testing.assert_array_equal(out.init_weights, z["forecasts_init_weights"])
    np.testing.assert_array_equal(out.samples_by_stage, z["forecasts_reshaped"])
    np.testing.assert_array_equal(out.goal_xy[:, 0], z["gx"])
    np.testing.assert_array_equal(out.goal_xy[:, 1], z["gy"])
    np.testing.assert_array_equal(out.v_pref, z["v_pref"])
    np.testing.assert_array_equal(human_headings(z["human_pxpyvxvy"][:, 2], z["human_pxpyvxvy"][:, 3]), z["theta"])
    p = stage_parameter_blocks(out.samples_by_stage, z["goal_states"], z["goal_actions"], z["Q_diag"], z["R_diag"],
                               z["term_Q_diag"], horiz, static_obs=z["static_obs"] if outdoor else None)
    assert p.shape == z["p_stages"].shape and p.dtype == np.float64
    np.testing.assert_array_equal(p, z["p_stages"])
    with pytest.raises(IndexError):          # the reference indexes MID_samples[horiz] and fails the same way
        stage_parameter_blocks(out.samples_by_stage[:horiz], z["goal_states"], z["goal_actions"], z["Q_diag"],
                               z["R_diag"], z["term_Q_diag"], horiz)


def test_install_aliases_the_reference_module(monkeypatch):
    """``install()``: the reference's caller line (sicnav_acados.py:24) resolves to this package's class without an edit -
    with the reference tree importable (its own parent packages) and without it (stand-in parents)."""
    import importlib
    import sys
    import safe_interactive_crowdnav_amd as P
    from safe_interactive_crowdnav_amd import forecaster as FC

    saved = dict(FC.DEFAULTS)
    for with_reference in (False, True):
        if with_reference and not os.path.isdir("/root/reference/sicnav_diffusion"):
            continue
        for name in [n for n in sys.modules if n == "sicnav_diffusion" or n.startswith("sicnav_diffusion.")]:
            monkeypatch.delitem(sys.modules, name)
        if with_reference:
            monkeypatch.syspath_prepend("/root/reference")
            importlib.invalidate_caches()
        try:
            assert FC.DEFAULTS["precision"] == "f16mx" and FC.DEFAULTS["self_check"] is True      # what bench.py quotes is what a drop-in user runs
            mod = P.install(precision="f16x3", self_check=False)
            stood_in = list(P._STAND_INS)
            assert mod is FC and FC.DEFAULTS["precision"] == "f16x3" and FC.DEFAULTS["self_check"] is False
            ns = {}
            exec("from sicnav_diffusion.JMID.mid_sim_wrapper import HumanTrajectoryForecasterSim", ns)     # the caller's line
            assert ns["HumanTrajectoryForecasterSim"] is FC.HumanTrajectoryForecasterSim
            import sicnav_diffusion.JMID.mid_sim_wrapper as W
            assert W is FC and hasattr(W, "get_most_likely_samples") and hasattr(W, "ForecasterSimSuper")
            with pytest.raises(TypeError):
                P.install(no_such_default=1)
        finally:
            P.uninstall()
            FC.DEFAULTS.clear()
            FC.DEFAULTS.update(saved)
        assert sys.modules.get("sicnav_diffusion.JMID.mid_sim_wrapper") is not FC
        # stand-in parents (registered when the reference tree is not importable) are gone again: the real packages stay
        # importable afterwards
        assert not P._STAND_INS and not any(n in sys.modules for n in stood_in)
        if with_reference:
            assert not stood_in


def test_install_does_not_mask_a_reference_package_that_fails_to_import(tmp_path, monkeypatch):
    """A reference tree that IS on sys.path but whose package __init__ raises (a missing dependency) must surface as that error:
    install() only substitutes a stand-in for "this package does not exist"."""
    import importlib
    import sys
    import safe_interactive_crowdnav_amd as P
    from safe_interactive_crowdnav_amd import forecaster as FC
    (tmp_path / "sicnav_diffusion").mkdir()
    (tmp_path / "sicnav_diffusion" / "__init__.py").write_text("import a_dependency_that_is_not_installed_here\n")
    for name in [n for n in sys.modules if n == "sicnav_diffusion" or n.startswith("sicnav_diffusion.")]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.syspath_prepend(str(tmp_path))
    importlib.invalidate_caches()
    saved = dict(FC.DEFAULTS)
    try:
        with pytest.raises(ModuleNotFoundError, match="a_dependency_that_is_not_installed_here"):
            P.install()
    finally:
        P.uninstall()
        FC.DEFAULTS.clear()
        FC.DEFAULTS.update(saved)
    assert "sicnav_diffusion" not in sys.modules


def test_chunk_plan_never_holds_an_empty_chunk():
    """run_network's chunk plan (csrc/jmid_planner.hip::plan_chunks, through the diagnostics entry point; host logic only): the
    sizes add up to E, no chunk is empty - round 4's balanced plan rounded E chunks of ONE episode (K*A*T > 32768 tokens per
    episode) up to E + 1 for odd E, and a zero-episode chunk is a grid-0 launch - and, unforced with two lanes, the count is even
    whenever that is possible."""
    import ctypes as C
    import __graft_entry__ as graft
    from safe_interactive_crowdnav_amd import _lib
    from safe_interactive_crowdnav_amd.build import LIB_DIAG
    graft.build()
    lib = _lib.load_library(LIB_DIAG)
    buf = (C.c_int * 8192)()

    def plan(E, tokens, lanes=2, forced=0, net=_lib.NET_JMID):
        n = lib.jmid_dbg_plan_chunks(net, 4, lanes, forced, E, tokens, buf, len(buf))
        assert 0 < n <= len(buf), n
        return list(buf[:n])

    for tokens in (1200, 19200, 38400, 128 * 25 * 12, 65536, 70000):       # cfg2, cfg4, and shapes whose automatic chunk is 1 episode
        for E in (1, 2, 3, 4, 5, 7, 9, 51, 52, 101, 255, 256, 257, 511, 512, 4096):
            for lanes in (1, 2, 3):
                for net in (_lib.NET_JMID, _lib.NET_IMID):
                    sizes = plan(E, tokens, lanes=lanes, net=net)
                    assert sum(sizes) == E and min(sizes) >= 1, (tokens, E, lanes, sizes[:8])
                    if lanes >= 2 and E >= 2 and len(sizes) < E:
                        assert len(sizes) % 2 == 0 and max(sizes) - min(sizes) <= 1, (tokens, E, sizes[:8])
            for forced in (1, 2, 51):
                sizes = plan(E, tokens, forced=forced)
                assert sum(sizes) == E and min(sizes) >= 1 and max(sizes) <= forced
    assert plan(256, 1200) == [43] * 4 + [42] * 2
    assert plan(3, 38400) == [1, 1, 1] and plan(5, 70000) == [1] * 5       # the advisor's case: c == 1, odd E

    # a small batch by arithmetic mode: at most 2 560 tokens stay ONE chunk in F16MX (the LayerNorms ride in its GEMM launches, which only
    # run while nothing else of the handle is in flight); every other mode, and anything larger, runs as two halves side by side
    def plan_mode(E, tokens, prec, lanes=2):
        n = lib.jmid_dbg_plan_chunks_mode(_lib.NET_JMID, 4, lanes, 0, E, tokens, prec, buf, len(buf))
        assert 0 < n <= len(buf), n
        return list(buf[:n])

    assert plan_mode(2, 1200, _lib.PREC_F16MX) == [2] and plan_mode(2, 1280, _lib.PREC_F16MX) == [2]
    assert plan_mode(2, 1281, _lib.PREC_F16MX) == [1, 1] and plan_mode(3, 1200, _lib.PREC_F16MX) == [2, 1]
    assert plan_mode(4, 600, _lib.PREC_F16MX) == [4] and plan_mode(4, 1200, _lib.PREC_F16MX) == [2, 2]
    for prec in (_lib.PREC_F32, _lib.PREC_F16X3, _lib.PREC_F16X2):
        assert plan_mode(2, 1200, prec) == [1, 1] and plan_mode(4, 600, prec) == [2, 2]
    assert plan_mode(2, 1200, _lib.PREC_F16MX, lanes=1) == [2] and plan_mode(256, 1200, _lib.PREC_F16MX) == [43] * 4 + [42] * 2
    for E in (1, 2, 3, 5, 51, 256):                      # the mode-less entry point = a mode without the rule
        assert plan(E, 1200) == plan_mode(E, 1200, _lib.PREC_F16X3)


def test_module_level_get_most_likely_samples_has_the_reference_signature():
    """mid_sim_wrapper.get_most_likely_samples(forecasts, mid_model, num_ret_samples) -> torch tensors [A, k, H, 2], [A, k]."""
    from safe_interactive_crowdnav_amd.forecaster import get_most_likely_samples, topk_fits_device
    z = np.load(os.path.join(GOLDEN, sorted(glob.glob(os.path.join(GOLDEN, "kde_*.npz")))[0]))
    top, lw = get_most_likely_samples(torch.from_numpy(z["forecasts"]), object(), int(z["k_ret"]))
    assert isinstance(top, torch.Tensor) and isinstance(lw, torch.Tensor)
    np.testing.assert_array_equal(top.numpy(), z["top"])
    np.testing.assert_allclose(lw.numpy(), z["logw"], rtol=1e-5, atol=1e-5)
    # the device kernel's size limits (include/jmid_hip.h): beyond them the class falls back to the host twin
    assert topk_fits_device(32, 1024, 24) and not topk_fits_device(33, 100, 8) and not topk_fits_device(5, 1025, 8) \
        and not topk_fits_device(5, 100, 25)


def test_rank_cpu_shares_are_disjoint_and_numa_local():
    """bench.rank_cpu_share (the per-rank CPU placement of a multi-rank launch): contiguous disjoint shares of the allowed cores; with
    NUMA information the ranks whose GPUs hang off one node split THAT node's cores; degenerate inputs never give an empty share."""
    import bench
    allowed = list(range(256))
    sh = [bench.rank_cpu_share(allowed, r, 8) for r in range(8)]
    assert all(len(x) == 32 for x in sh) and sorted(sum(sh, [])) == allowed
    numa = [0, 0, 0, 0, 1, 1, 1, 1]
    nodes = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    sh = [bench.rank_cpu_share(allowed, r, 8, numa, nodes) for r in range(8)]
    assert sorted(sum(sh, [])) == allowed and all(set(sh[r]) <= set(nodes[numa[r]]) for r in range(8))
    # a cpuset smaller than the node list (a container): only allowed cores are handed out
    sh = [bench.rank_cpu_share(list(range(0, 16)), r, 8, numa, nodes) for r in range(8)]
    assert all(len(x) >= 1 and set(x) <= set(range(16)) for x in sh)
    assert bench.rank_cpu_share([3, 4], 5, 8) == [3, 4]                      # fewer cores than ranks: everybody gets them all
    assert bench.rank_cpu_share(allowed, 2, 3, [None, None, None], {}) == allowed[171:256]
    assert bench._cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]


def test_compact_bench_line_always_fits_and_always_parses():
    """bench.compact_line never asserts and never loses the contract keys, however long the notes of the long result are (the driver
    parses this one stdout line; round 4's 20 KB line was not parsed)."""
    import json
    import bench
    full = {"metric": "m", "value": 1.0, "unit": "traj/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "x" * 3000, "data": "synthetic",
            "config": {"workload_short": "w" * 3000, "episodes_per_gpu": 1, "total_episodes": 1, "humans": 5, "samples": 20, "horizon": 12,
                       "denoise_steps": 50, "net": "jmid", "precision": "f16mx", "lanes": 2, "dist_backend": None},
            "cpu_baseline": {"value": 1.0, "unit": "traj/s", "cores": 1, "kind": "port", "processes": 1, "threads_per_process": 1,
                             "hardware_threads": 1, "host_physical_cores": 1, "sample_short": "s" * 3000, "cores_short": "c" * 3000},
            "modes": {m: {"value": 1.0, "ms_per_step": 1.0} for m in ("f16mx", "f16x2", "f16x3")}}
    line = bench.compact_line(full, "/tmp/detail.json")
    assert len(line) <= bench.COMPACT_LIMIT
    j = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline"):
        assert k in j, k
