"""Pins oracle/jmid_oracle.py against outputs of the REFERENCE (tests/golden/*.npz, produced by
tests/golden/make_golden.py importing /root/reference in the build container)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import jmid_oracle as O
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _weights(z):
    w = JMIDWeights.from_seed(NetDims(ctx_dim=int(z["ctx_dim"])), int(z["wseed"]))
    assert w.checksum() == str(z["wsum"]), "synthetic weights differ from the ones the reference was run with"
    return w


def test_schedule_matches_reference():
    z = np.load(os.path.join(GOLDEN, "schedule.npz"))
    s = O.variance_schedule()
    for k in ("betas", "alphas", "alpha_bars", "sigmas_flex", "sigmas_inflex"):
        np.testing.assert_array_equal(s[k].numpy(), z[k], err_msg=k)
    # spot values recorded in SURVEY.md 8(a6)
    assert s["betas"][1].item() == pytest.approx(9.9999997e-05, rel=1e-7)
    assert s["alpha_bars"][100].item() == pytest.approx(0.078234285, rel=1e-6)


NET_CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "net_*.npz")))


@pytest.mark.parametrize("case", NET_CASES)
def test_net_and_sampler_match_reference(case):
    z = np.load(os.path.join(GOLDEN, case))
    w = _weights(z).tensors
    A, K, T, step, joint = int(z["A"]), int(z["K"]), int(z["T"]), int(z["step"]), bool(z["joint"])
    ctx, x_T = torch.from_numpy(z["ctx"]), torch.from_numpy(z["x_T"])
    torch.set_num_threads(8)
    # the cfg4 fixture (one 19 200-token sequence, 50 steps) is ~4 minutes of oracle time on 8 cores: its sampler loop runs
    # with JMID_SLOW_TESTS=1 (passed, 2e-6, when the fixture was added); by default only its first net evaluation is checked.
    # The HIP path is held against this fixture directly on the GPU (tests/test_gpu_baseline_sizes.py).
    heavy = K * A * T * step > 500_000 and not os.environ.get("JMID_SLOW_TESTS")
    with torch.no_grad():
        beta = O.variance_schedule()["betas"][[100] * (K * A)]
        e = O.net_forward(w, x_T, ctx.repeat(K, 1), beta, joint=joint)
        vel = None if heavy else O.denoise(w, ctx, x_T, sample=K, step=step, joint=joint)
    # mean ADE-style metric: mean L2 over (sample, agent, t).  Gate of the project is 1e-4.
    # The oracle restates the same torch-CPU ops: bit-exact at width 32, and within fp32 GEMM
    # blocking noise at width 256 (measured <= 1.4e-6 after 50 steps, below the reference's own
    # fp32-vs-fp64 distance of 1.7e-6) -> held to 5e-6.
    d_e = np.linalg.norm(e.numpy() - z["e_first"], axis=-1).mean()
    assert d_e <= 1e-6, d_e
    assert int(z["nsteps"]) == K * (100 // int(100 / step) + 1)
    if heavy:
        pytest.skip("first net evaluation checked; the 50-step sampler of this fixture runs with JMID_SLOW_TESTS=1")
    d_v = np.linalg.norm(vel.numpy() - z["vel"], axis=-1).mean()
    assert d_v <= 5e-6, d_v


WRAP_CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "wrapper_*.npz")))


@pytest.mark.parametrize("case", WRAP_CASES)
def test_context_encoder_matches_reference(case):
    z = np.load(os.path.join(GOLDEN, case))
    w = _weights(z).tensors
    with torch.no_grad():
        ctx = O.encode_context(w, torch.from_numpy(z["x_st"]), torch.from_numpy(z["nbr_sum"]),
                               torch.from_numpy(z["edge_mask"]))
    np.testing.assert_allclose(ctx.numpy(), z["ctx"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("case", WRAP_CASES)
def test_sampler_and_integrator_on_wrapper_capture(case):
    z = np.load(os.path.join(GOLDEN, case))
    w = _weights(z).tensors
    K, H, step, joint = int(z["K"]), int(z["H"]), int(z["step"]), bool(z["joint"])
    ctx = torch.from_numpy(z["ctx"])
    A = ctx.shape[0]
    torch.manual_seed(int(z["dseed"]))
    x_T = torch.randn([K * A, H, 2])  # first draw after seeding (diffusion.py:499)
    with torch.no_grad():
        vel = O.denoise(w, ctx, x_T, sample=K, step=step, joint=joint)
    assert np.linalg.norm(vel.numpy() - z["vel"], axis=-1).mean() <= 5e-6
    # integrator + scatter: in-cluster rows of the final forecasts (t0 prepended) when no top-k selection
    if int(z["k_ret"]) >= K:
        p0 = torch.from_numpy(z["x_t"][:, -1, 0:2])
        pos = O.integrate(vel, p0, float(z["time_step"]))           # [K,A,H,2]
        order = np.argsort(z["node_ids"])
        pos = pos.numpy()[:, order].transpose(1, 0, 2, 3)            # [A,K,H,2] sorted by track id
        ids = np.sort(z["node_ids"])
        np.testing.assert_allclose(z["forecasts"][ids][:, :, 1:, :], pos, rtol=0, atol=1e-5)


@pytest.mark.parametrize("case", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "kde_*.npz"))))
def test_kde_topk_matches_reference(case):
    z = np.load(os.path.join(GOLDEN, case))
    top, lw = O.most_likely_samples(torch.from_numpy(z["forecasts"]), int(z["k_ret"]))
    np.testing.assert_allclose(top.numpy(), z["top"], rtol=0, atol=0)
    np.testing.assert_allclose(lw.numpy(), z["logw"], rtol=1e-5, atol=1e-5)


DDPM_CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "ddpm_*.npz")))


@pytest.mark.parametrize("case", DDPM_CASES)
def test_ddpm_sampling_matches_reference(case):
    z = np.load(os.path.join(GOLDEN, case))
    w = _weights(z).tensors
    with torch.no_grad():
        vel = O.denoise(w, torch.from_numpy(z["ctx"]), torch.from_numpy(z["x_T"]), sample=int(z["K"]), step=int(z["step"]),
                        joint=bool(z["joint"]), sampling="ddpm", z=torch.from_numpy(z["z"]))
    assert np.linalg.norm(vel.numpy() - z["vel"], axis=-1).mean() <= 5e-6


SAMPLE_CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "sample_*.npz")))


@pytest.mark.parametrize("case", SAMPLE_CASES)
def test_offline_sample_matches_reference(case):
    """DiffusionTraj.sample (diffusion.py:544-613): per-sample loop, global-generator draws, flexibility."""
    z = np.load(os.path.join(GOLDEN, case))
    w = _weights(z).tensors
    torch.manual_seed(int(z["dseed"]))
    with torch.no_grad():
        vel = O.sample_offline(w, torch.from_numpy(z["ctx"]), int(z["T"]), int(z["n_sample"]), bool(z["bestof"]),
                               step=int(z["step"]), joint=bool(z["joint"]), sampling=str(z["sampling"]),
                               flexibility=float(z["flexibility"]))
    assert vel.shape == z["vel"].shape
    assert np.linalg.norm(vel.numpy() - z["vel"], axis=-1).mean() <= 5e-6
