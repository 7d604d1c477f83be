"""Parity of the HIP path (through the C ABI) with the REFERENCE golden vectors and with the oracle.

mean ADE = mean over (episode, sample, agent, t) of the L2 distance, the per-element definition of
compute_ade (MID/evaluation/evaluation.py:20-26) applied between two predictors.  Gate: 1e-4 (north_star).
"""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import jmid_oracle as O
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PROD_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "safe-interactive-crowdnav_amd", "csrc", "libjmid_hip.so")
ADE_GATE = 1e-4
PRECISIONS = ["f32", "f16x3", "f16x2", "f16mx"]
SPLIT_MODES = ["f16x3", "f16x2", "f16mx"]      # all run on the hi/lo operand planes; f16x2 leaves the activation-lo term out,
# f16mx = f16x2 with the weight-lo correction term of every GEMM as one bf8 x bf8 MFMA per k64
# one e_theta evaluation: fp32-level for f32 / f16x3; f16x2 rounds each linear layer's input to fp16 (2^-12 relative)
E_THETA_TOL = {"f32": 1e-5, "f16x3": 1e-5, "f16x2": 5e-4, "f16mx": 5e-4}


def ade(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64), axis=-1).mean())


_ENGINES = {}


def get_engine(ctx_dim, wseed, joint, flavour="diag"):
    """`flavour`: "diag" = the diagnostics build the test session loads (tests/conftest.py), "prod" = the PRODUCTION
    libjmid_hip.so (what the drop-in class, bench.py and smoke() load), side by side in this process."""
    key = (ctx_dim, wseed, joint, flavour)
    if key not in _ENGINES:
        w = JMIDWeights.from_seed(NetDims(ctx_dim=ctx_dim), wseed)
        _ENGINES[key] = (JmidEngine(w, joint=joint, lib_path={"prod": PROD_LIB}.get(flavour)), w)
        assert _ENGINES[key][0]._lib.has_diagnostics == (flavour != "prod")
    return _ENGINES[key]


NET_CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "net_*.npz")))


@pytest.mark.parametrize("flavour", ["diag", "prod"])
@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", NET_CASES)
def test_net_eval_and_denoise_match_reference_golden(case, precision, flavour):
    """Every reference capture of the net / the DDIM loop, every mode, on BOTH builds of the library: the diagnostics flavour the
    rest of this file drives and the production flavour that ships."""
    z = np.load(os.path.join(GOLDEN, case))
    eng, w = get_engine(int(z["ctx_dim"]), int(z["wseed"]), bool(z["joint"]), flavour)
    assert w.checksum() == str(z["wsum"])
    A, K, T, step = int(z["A"]), int(z["K"]), int(z["T"]), int(z["step"])
    eng.set_step(step)
    ctx = z["ctx"][None]                 # [1, A, C]
    x_T = z["x_T"][None]                 # [1, K*A, T, 2]
    e = eng.net_eval(x_T, ctx, step_idx=0, precision=precision)[0]
    assert ade(e, z["e_first"]) <= E_THETA_TOL[precision], ("e_theta", ade(e, z["e_first"]))
    vel, _ = eng.denoise(x_T, ctx, precision=precision, want_pos=False)
    a = ade(vel[0], z["vel"])
    print(f"{case} [{precision}] mean ADE(vel) vs reference = {a:.3e}")
    assert a <= ADE_GATE, a


WRAP_CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "wrapper_*.npz")))


@pytest.mark.parametrize("case", WRAP_CASES)
def test_context_encoder_matches_reference_golden(case):
    z = np.load(os.path.join(GOLDEN, case))
    eng, w = get_engine(int(z["ctx_dim"]), int(z["wseed"]), bool(z["joint"]))
    ctx = eng.encode(z["x_st"], z["nbr_sum"], z["edge_mask"])
    np.testing.assert_allclose(ctx, z["ctx"], rtol=0, atol=5e-6)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("joint", [True, False])
def test_multi_episode_batch_is_block_diagonal(joint, precision):
    """E episodes in one call == E independent single-scene calls == the oracle per episode,
    for any chunking (JMID attention must not leak across episodes)."""
    eng, w = get_engine(32, 77, joint)
    eng.set_step(10)
    E, A, K, T = 5, 3, 4, 6
    g = torch.Generator().manual_seed(5)
    ctx = torch.randn([E, A, 32], generator=g)
    x_T = torch.randn([E, K * A, T, 2], generator=g)
    p0 = torch.randn([E, A, 2], generator=g)
    with torch.no_grad():
        ref = torch.stack([O.denoise(w.tensors, ctx[e], x_T[e], sample=K, step=10, joint=joint) for e in range(E)])
        ref_multi = O.denoise(w.tensors, ctx, x_T, sample=K, step=10, joint=joint)
        ref_pos = O.integrate(ref, p0, 0.25)
    assert ade(ref_multi.numpy(), ref.numpy()) <= 1e-6  # the oracle's own multi-episode extension
    outs = []
    for chunk in (0, 1, 2, 5):
        eng.set_chunk_episodes(chunk)
        vel, pos = eng.denoise(x_T.numpy(), ctx.numpy(), p0.numpy(), dt=0.25, precision=precision)
        assert ade(vel, ref.numpy()) <= ADE_GATE
        assert ade(pos, ref_pos.numpy()) <= ADE_GATE
        outs.append(vel)
    eng.set_chunk_episodes(0)
    for o in outs[1:]:
        np.testing.assert_array_equal(o, outs[0])  # chunking must not change a single bit


def test_device_pointers_match_host_path():
    eng, w = get_engine(32, 77, True)
    eng.set_step(5)
    E, A, K, T = 2, 2, 3, 4
    g = torch.Generator().manual_seed(9)
    ctx = torch.randn([E, A, 32], generator=g)
    x_T = torch.randn([E, K * A, T, 2], generator=g)
    p0 = torch.randn([E, A, 2], generator=g)
    vel_h, pos_h = eng.denoise(x_T.numpy(), ctx.numpy(), p0.numpy())
    vel_d, pos_d = eng.denoise(x_T.cuda(), ctx.cuda(), p0.cuda())
    eng.synchronize()
    np.testing.assert_array_equal(vel_d.cpu().numpy(), vel_h)
    np.testing.assert_array_equal(pos_d.cpu().numpy(), pos_h)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_full_size_property_checks(precision):
    """BASELINE cfg3-like size (scaled: 32 episodes x N=5 x K=20 x H=12, 50 steps) through size-independent
    properties: (a) permuting episodes permutes outputs (bit-exactly except for rows that move between an interior
    and an edge tile of a GEMM launch, where the epilogue code path differs and results may move by an ulp or two:
    held to 1e-5 m), (b) duplicated episodes in one batch give identical outputs and reruns are deterministic,
    (c) pos = cumsum(vel)*dt + p0."""
    eng, w = get_engine(256, 23, True)
    eng.set_step(50)
    E, A, K, T = 32, 5, 20, 12
    g = torch.Generator().manual_seed(3)
    ctx = torch.randn([E, A, 256], generator=g)
    x_T = torch.randn([E, K * A, T, 2], generator=g)
    p0 = torch.randn([E, A, 2], generator=g)
    ctx[7], x_T[7], p0[7] = ctx[3], x_T[3], p0[3]
    vel, pos = eng.denoise(x_T.cuda(), ctx.cuda(), p0.cuda(), dt=0.25, precision=precision)
    vel, pos = vel.cpu(), pos.cpu()
    assert torch.isfinite(vel).all()
    assert torch.equal(vel[7], vel[3]) and torch.equal(pos[7], pos[3])
    perm = torch.randperm(E, generator=g)
    vel_p, _ = eng.denoise(x_T[perm].cuda(), ctx[perm].cuda(), p0[perm].cuda(), dt=0.25, precision=precision)
    dperm = (vel_p.cpu() - vel[perm]).norm(dim=-1)
    assert dperm.mean() <= 1e-6 and dperm.max() <= 1e-3     # ulp-level seeds amplified over 50 chaotic steps
    vel_r, _ = eng.denoise(x_T[perm].cuda(), ctx[perm].cuda(), p0[perm].cuda(), dt=0.25, precision=precision)
    assert torch.equal(vel_r.cpu(), vel_p.cpu())      # rerun determinism
    ref_pos = torch.cumsum(vel, dim=3) * 0.25 + p0[:, None, :, None, :]
    assert (pos - ref_pos).abs().max() <= 1e-4


@pytest.mark.expects_erange
def test_f16x3_reports_range_overflow_instead_of_garbage():
    """JMID_PREC_F16X3 carries operands as fp16 hi/lo planes: values beyond the fp16 range must surface as
    JMID_ERANGE (the caller then reruns in JMID_PREC_F32), never as silent inf/NaN trajectories - and are counted on the
    handle (jmid_erange_count) so that a deployment can see how often the slow path fires."""
    from safe_interactive_crowdnav_amd.engine import JmidError
    eng, w = get_engine(32, 77, True)
    eng.set_step(2)
    n0 = eng.erange_count()
    g = torch.Generator().manual_seed(1)
    ctx = torch.randn([1, 3, 32], generator=g)
    x_T = torch.randn([1, 6, 4, 2], generator=g) * 1e9
    with pytest.raises(JmidError) as ei:
        eng.denoise(x_T.numpy(), ctx.numpy(), precision="f16x3", want_pos=False)
    assert ei.value.code == -5
    assert eng.erange_count() == n0 + 1
    vel, _ = eng.denoise(x_T.numpy(), ctx.numpy(), precision="f32", want_pos=False)   # fp32 path still answers
    assert np.isfinite(vel).all()
    assert eng.erange_count() == n0 + 1


@pytest.mark.parametrize("precision", PRECISIONS)
def test_dense_crowd_shape_matches_oracle(precision):
    """BASELINE configs[3] geometry (N=25 humans) at a sample count the oracle finishes quickly (K=6 -> one JMID
    sequence of 1800 tokens), ragged against every tile size (1800 = 14*128 + 8), 5 DDIM steps."""
    eng, w = get_engine(256, 23, True)
    eng.set_step(5)
    E, A, K, T = 2, 25, 6, 12
    g = torch.Generator().manual_seed(11)
    ctx = torch.randn([E, A, 256], generator=g)
    x_T = torch.randn([E, K * A, T, 2], generator=g)
    with torch.no_grad():
        ref = O.denoise(w.tensors, ctx, x_T, sample=K, step=5, joint=True)
    vel, _ = eng.denoise(x_T.numpy(), ctx.numpy(), precision=precision, want_pos=False)
    a = ade(vel, ref.numpy())
    print(f"dense crowd [{precision}] mean ADE vs oracle = {a:.3e}")
    assert a <= ADE_GATE


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "ddpm_*.npz"))))
def test_ddpm_sampling_matches_reference_golden(case, precision):
    """sampling="ddpm" (MID/models/diffusion.py:509-522): x <- c0 (x - c1 e) + sigma z with the reference's z draws."""
    z = np.load(os.path.join(GOLDEN, case))
    eng, w = get_engine(int(z["ctx_dim"]), int(z["wseed"]), bool(z["joint"]))
    assert w.checksum() == str(z["wsum"])
    eng.set_step(int(z["step"]), "ddpm")
    try:
        vel, _ = eng.denoise(z["x_T"][None], z["ctx"][None], precision=precision, want_pos=False, z=z["z"][:, None])
        with pytest.raises(Exception):
            eng.denoise(z["x_T"][None], z["ctx"][None], precision=precision, want_pos=False)   # DDPM needs z
    finally:
        eng.set_step(int(z["step"]), "ddim")
    a = ade(vel[0], z["vel"])
    print(f"{case} [{precision}] DDPM mean ADE vs reference = {a:.3e}")
    assert a <= ADE_GATE


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "sample_*.npz"))))
def test_offline_sample_matches_reference_golden(case, precision):
    """offline.sample == DiffusionTraj.sample (MID/models/diffusion.py:544-613) on the reference's RNG draws."""
    from safe_interactive_crowdnav_amd import offline
    z = np.load(os.path.join(GOLDEN, case))
    eng, w = get_engine(int(z["ctx_dim"]), int(z["wseed"]), bool(z["joint"]))
    assert w.checksum() == str(z["wsum"])
    step0, samp0 = eng.step, eng.sampling
    torch.manual_seed(int(z["dseed"]))
    try:
        vel, nsteps, a, b, c = offline.sample(eng, int(z["T"]), z["ctx"], int(z["n_sample"]), bool(z["bestof"]),
                                              flexibility=float(z["flexibility"]), sampling=str(z["sampling"]),
                                              step=int(z["step"]), precision=precision)
    finally:
        eng.set_step(step0, samp0)
    assert vel.shape == z["vel"].shape and nsteps == int(z["nsteps"]) and (a, b, c) == (0, 0, 0)
    d = ade(vel, z["vel"])
    print(f"{case} [{precision}] offline sample mean ADE vs reference = {d:.3e}")
    assert d <= ADE_GATE


def test_offline_generate_integrates_on_the_device():
    """offline.generate = sample + SingleIntegrator.integrate_samples (single_integrator.py:290-321): the positions come from
    integrate_kernel in the same call and equal dt * cumsum(vel) + p0 of offline.sample on the same RNG draws."""
    from safe_interactive_crowdnav_amd import offline
    case = sorted(glob.glob(os.path.join(GOLDEN, "sample_*.npz")))[0]
    z = np.load(case)
    eng, w = get_engine(int(z["ctx_dim"]), int(z["wseed"]), bool(z["joint"]))
    step0, samp0 = eng.step, eng.sampling
    B = z["ctx"].shape[0]
    p0 = np.random.default_rng(4).standard_normal((B, 2)).astype(np.float32)
    kw = dict(flexibility=float(z["flexibility"]), sampling=str(z["sampling"]), step=int(z["step"]), precision="f32")
    try:
        torch.manual_seed(int(z["dseed"]))
        vel = offline.sample(eng, int(z["T"]), z["ctx"], int(z["n_sample"]), bool(z["bestof"]), **kw)[0]
        torch.manual_seed(int(z["dseed"]))
        pos, nsteps, a, b, c = offline.generate(eng, z["ctx"], p0, 0.4, int(z["T"]), int(z["n_sample"]), bool(z["bestof"]), **kw)
    finally:
        eng.set_step(step0, samp0)
    assert pos.shape == vel.shape and pos.dtype == np.float32 and nsteps == int(z["nsteps"])
    ref = np.cumsum(vel.astype(np.float64), axis=2) * 0.4 + p0[None, :, None, :]
    np.testing.assert_allclose(pos, ref, rtol=0, atol=2e-6 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("precision", SPLIT_MODES)
@pytest.mark.parametrize("case", ["net_jmid_w256_a5k20t12_s50.npz", "net_jmid_w32_a5k20t12_s50.npz"])
def test_fused_vt_epilogue_equals_transpose_kernel(case, precision):
    """The QKV epilogue that writes V^T itself (S % 4 == 0) and the row-major V + v_transpose_kernel path hold the
    same values: bit-identical trajectories, and both at reference parity."""
    z = np.load(os.path.join(GOLDEN, case))
    eng, _ = get_engine(int(z["ctx_dim"]), int(z["wseed"]), True)
    eng.set_step(int(z["step"]))
    out = []
    try:
        for off in (0, 1):
            eng.set_tuning("no_vt_direct", off)
            out.append(eng.denoise(z["x_T"][None], z["ctx"][None], precision=precision, want_pos=False)[0][0])
    finally:
        eng.set_tuning("no_vt_direct", 0)
    np.testing.assert_array_equal(out[0], out[1])
    assert ade(out[0], z["vel"]) <= ADE_GATE


@pytest.mark.parametrize("case", ["net_jmid_w256_a5k20t12_s50.npz", "net_imid_w256_a5k20t12_s50.npz",
                                  "net_jmid_w256_a7k9t24_s10.npz"])
@pytest.mark.parametrize("precision", SPLIT_MODES)
def test_fused_gemm_layernorm_equals_gemm_then_add_ln(case, precision):
    """gemm_ln_f16x3_kernel (row-complete GEMM + residual + LayerNorm) and GEMM -> fp32 Y -> add_ln do the same
    arithmetic in the same order: bit-identical trajectories, whichever the token count selects."""
    z = np.load(os.path.join(GOLDEN, case))
    eng, _ = get_engine(int(z["ctx_dim"]), int(z["wseed"]), bool(z["joint"]))
    eng.set_step(int(z["step"]))
    out = {}
    try:
        for mode, rows in ((1, 64), (1, 128), (2, 0)):      # fused with 64- / 128-row tiles, never fused
            eng.set_tuning("ln_fuse", mode)
            eng.set_tuning("ln_rows", rows)
            out[(mode, rows)] = eng.denoise(z["x_T"][None], z["ctx"][None], precision=precision, want_pos=False)[0][0]
    finally:
        eng.set_tuning("ln_fuse", 0)
        eng.set_tuning("ln_rows", 0)
    np.testing.assert_array_equal(out[(1, 64)], out[(2, 0)])
    np.testing.assert_array_equal(out[(1, 128)], out[(2, 0)])
    assert ade(out[(1, 64)], z["vel"]) <= ADE_GATE


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", ["net_imid_w256_a5k20t12_s50.npz", "net_imid_w32_a2k3t4_s50.npz"])
def test_packed_short_sequence_attention_matches_unpacked(case, precision):
    """iMID sequences (S = T <= 16) share a wave's score tile (attn_f32_packed_kernel); one sequence per wave is the
    old path.  Same math, different summation slots: both at reference parity and within 1e-5 of each other - 2e-5 in the modes that
    round P to one fp16 plane (f16x2 / f16mx), where since round 6 the one-sequence-per-wave kernel takes the row sum from the rounded
    P and the packed kernel from the fp32 values (50 steps carry that last-bit difference to ~1e-5 m)."""
    z = np.load(os.path.join(GOLDEN, case))
    eng, _ = get_engine(int(z["ctx_dim"]), int(z["wseed"]), False)
    eng.set_step(int(z["step"]))
    out = {}
    try:
        for mode in (1, 0):
            eng.set_tuning("attn_pack", mode)
            out[mode] = eng.denoise(z["x_T"][None], z["ctx"][None], precision=precision, want_pos=False)[0][0]
    finally:
        eng.set_tuning("attn_pack", 1)
    assert ade(out[1], z["vel"]) <= ADE_GATE and ade(out[0], z["vel"]) <= ADE_GATE
    assert ade(out[1], out[0]) <= (2e-5 if precision in ("f16x2", "f16mx") else 1e-5)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("joint", [True, False])
@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (2, 3, 3, 5), (3, 2, 1, 24), (2, 5, 7, 3), (1, 4, 9, 17)])
def test_edge_shapes_match_oracle(shape, joint, precision):
    """Degenerate and ragged geometries at full width: a single token, odd sequence lengths (S % 4 != 0: the V^T
    transpose kernel instead of the fused epilogue; partial packed-attention tiles), K = 1, T = max_len 24, T = 17
    (iMID sequences longer than one packed slot)."""
    E, A, K, T = shape
    eng, w = get_engine(256, 23, joint)
    eng.set_step(5)
    g = torch.Generator().manual_seed(100 + E + 7 * A + 31 * K + 97 * T)
    ctx = torch.randn([E, A, 256], generator=g)
    x_T = torch.randn([E, K * A, T, 2], generator=g)
    with torch.no_grad():
        ref = O.denoise(w.tensors, ctx, x_T, sample=K, step=5, joint=joint)
    vel, _ = eng.denoise(x_T.numpy(), ctx.numpy(), precision=precision, want_pos=False)
    a = ade(vel, ref.numpy())
    print(f"shape {shape} joint={joint} [{precision}] mean ADE vs oracle = {a:.3e}")
    assert np.isfinite(vel).all() and a <= ADE_GATE


def test_chunk_lanes_are_bit_identical():
    """Several chunks in flight on separate streams (jmid_set_tuning "lanes", 2 by default) compute the same bits as one
    chunk in flight.  (In round 1 they did not: packed-fp32 instructions with crossed operand selects in the row-wise
    kernels went wrong next to the other lane's attention workgroups - the library is built without them now, DESIGN.md
    section 3; tests/test_abi.py checks the device code.)"""
    eng, w = get_engine(32, 77, True)
    eng.set_step(10)
    E, A, K, T = 7, 3, 4, 6
    g = torch.Generator().manual_seed(9)
    ctx = torch.randn([E, A, 32], generator=g).numpy()
    x_T = torch.randn([E, K * A, T, 2], generator=g).numpy()
    try:
        eng.set_chunk_episodes(2)            # 4 chunks, the last one ragged
        eng.set_tuning("lanes", 1)
        ref = eng.denoise(x_T, ctx, precision="f16x3", want_pos=False)[0]
        np.testing.assert_array_equal(eng.denoise(x_T, ctx, precision="f16x3", want_pos=False)[0], ref)
        for lanes in (2, 3, 4):
            eng.set_tuning("lanes", lanes)
            np.testing.assert_array_equal(eng.denoise(x_T, ctx, precision="f16x3", want_pos=False)[0], ref)
        # row-wise kernels kept off the CUs of the MFMA kernels by an LDS request they never use (round-1 workaround): same values
        eng.set_tuning("bystander_lds", 96 * 1024)
        for lanes in (1, 2):
            eng.set_tuning("lanes", lanes)
            np.testing.assert_array_equal(eng.denoise(x_T, ctx, precision="f16x3", want_pos=False)[0], ref)
        with pytest.raises(Exception):
            eng.set_tuning("bystander_lds", 1 << 20)
    finally:
        eng.set_tuning("bystander_lds", 0)
        eng.set_tuning("lanes", 2)
        eng.set_chunk_episodes(0)


def test_split_modes_hold_parity_when_attention_is_peaked():
    """Random-init weights give softmax logits of order 0.1; a trained network's are larger.  With the Q and K rows of
    every in_proj scaled by 8 (logits x 64) both split-fp16 modes must still sit inside the gate against the exact-fp32
    mode of the same library: f16x3 at fp32 level, f16x2 (whose logits keep all three product terms, exactly because a
    rounded logit is exponentiated) well inside 1e-4 m (measured 1.7e-5 m on the full cfg3 chunk)."""
    w = JMIDWeights.from_seed(NetDims(ctx_dim=256), 0)
    for k, t in w.tensors.items():
        if "in_proj" in k:
            t[: 2 * t.shape[0] // 3] *= 8.0
    eng = JmidEngine(w, joint=True, step=50)
    E, A, K, T = 6, 5, 20, 12
    g = torch.Generator().manual_seed(5)
    ctx = torch.randn([E, A, 256], generator=g).cuda()
    x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
    p0 = (4 * torch.randn([E, A, 2], generator=g)).cuda()
    vel32, ref = (t.cpu().numpy() for t in eng.denoise(x_T, ctx, p0, precision="f32"))
    out = {m: eng.denoise(x_T, ctx, p0, precision=m) for m in SPLIT_MODES}
    err = {m: ade(out[m][1].cpu().numpy(), ref) for m in SPLIT_MODES}
    print("peaked attention: mean ADE vs exact fp32", err)
    assert err["f16x3"] <= 1e-5, err
    assert err["f16x2"] <= ADE_GATE, err
    assert err["f16mx"] <= ADE_GATE, err
    # ... and against the ORACLE (not only the library's own fp32 mode) on the first episode: the large-logit case is the one a
    # trained network presents to the fp16 / bf8 logit corrections
    with torch.no_grad():
        oref = O.denoise(w.tensors, ctx[0].cpu(), x_T[0].cpu(), sample=K, step=50, joint=True).reshape(K * A, T, 2).numpy()
    oerr = {"f32": ade(vel32[0].reshape(K * A, T, 2), oref),
            **{m: ade(out[m][0][0].cpu().numpy().reshape(K * A, T, 2), oref) for m in SPLIT_MODES}}
    print("peaked attention: mean velocity ADE vs the oracle (episode 0)", oerr)
    assert oerr["f32"] <= 1e-5 and oerr["f16x3"] <= 1e-5, oerr
    assert oerr["f16x2"] <= ADE_GATE and oerr["f16mx"] <= ADE_GATE, oerr


@pytest.mark.parametrize("E,A,K,T,step", [(1, 3, 100, 8, 2), (1, 3, 100, 8, 50), (1, 5, 35, 12, 10), (1, 5, 68, 12, 4), (2, 5, 20, 12, 10)])
def test_layernorm_inside_the_gemm_launch_at_two_workgroups_per_cu_is_bit_identical(E, A, K, T, step):
    """A scene of 2 049 ... 4 096 tokens in F16MX (the reference's shipped point: N = 3, K = 100, H = 8 = 2 400 tokens): the one-launch
    GEMM + LayerNorm with the statistics exchange (and the split-KV merge in the out-projection's launch) runs with TWO workgroups per CU
    (gemm_small.hpp, SmCfg<SM_MX, 2, true>: 33 ... 64 row tiles x 8 = up to 512 workgroups, all resident).  "small_lnx2" = 2 takes the
    separate launches: the same bits, call after call on one handle, and within the gate of the oracle.  E = 2: two cfg2 scenes, which the
    planner keeps in ONE chunk for the sake of this kernel (with the knob at 2: two halves side by side - the same bits)."""
    eng, w = get_engine(256, 31, True)
    eng.set_step(step, "ddim")
    g = torch.Generator().manual_seed(A * 1000 + K)
    ctx = torch.randn([E, A, 256], generator=g).cuda()
    x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
    assert 2048 < E * K * A * T <= 4096
    out = []
    try:
        for knob in (2, 0, 0, 2, 0):
            eng.set_tuning("small_lnx2", knob)
            out.append(eng.denoise(x_T, ctx, precision="f16mx", want_pos=False)[0].cpu().numpy())
    finally:
        eng.set_tuning("small_lnx2", 0)
    for o in out[1:]:
        np.testing.assert_array_equal(o, out[0])
    if step <= 10:
        with torch.no_grad():
            ref = O.denoise(w.tensors, ctx.cpu(), x_T.cpu(), sample=K, step=step, joint=True).numpy()
        assert ade(out[1], ref) <= ADE_GATE


@pytest.mark.parametrize("case", ["net_jmid_w256_a5k20t12_s50.npz", "net_imid_w256_a5k20t12_s50.npz"])
def test_one_scene_layernorm_inside_the_gemm_launch_is_bit_identical(case):
    """One scene in F16MX (d_model 512): out_proj / linear2 + residual + LayerNorm run as ONE small launch whose workgroups exchange
    the row statistics ONCE - block sums and squared deviations from the block means, the mode's canonical order since round 6
    (gemm_small.hpp, OUT_LNX; six launches less per denoise step): the same bits as GEMM + add_ln2 over a whole 50-step loop, repeated on
    one handle (the exchange buffer is reused by every launch of every call); no workgroup ever gives up waiting (that would come back
    as JMID_ETIMEOUT, be retried by the engine and recorded)."""
    from safe_interactive_crowdnav_amd import engine as EN
    z = np.load(os.path.join(GOLDEN, case))
    eng, _ = get_engine(int(z["ctx_dim"]), int(z["wseed"]), bool(z["joint"]))
    eng.set_step(int(z["step"]), "ddim")
    out = []
    n0 = len(EN.TIMEOUT_EVENTS)
    try:
        for knob in (2, 0, 0, 2, 0):
            eng.set_tuning("small_lnx", knob)
            out.append(eng.denoise(z["x_T"][None], z["ctx"][None], precision="f16mx", want_pos=False)[0][0])
    finally:
        eng.set_tuning("small_lnx", 0)
    for o in out[1:]:
        np.testing.assert_array_equal(o, out[0])
    assert len(EN.TIMEOUT_EVENTS) == n0 and eng.timeout_count() == 0
    assert ade(out[3], z["vel"]) <= ADE_GATE


@pytest.mark.parametrize("precision", ["f16mx", "f16x2"])
@pytest.mark.parametrize("case", ["net_jmid_w256_a5k20t12_s50.npz", "net_jmid_w256_a7k9t24_s10.npz"])
def test_one_scene_split_kv_merge_inside_the_out_projection_launch_is_bit_identical(case, precision):
    """One scene: the attention launch splits the key range over several workgroups per query tile; their partial outputs are merged
    by the workgroups of the out-projection launch in front of their K loops (gemm_small.hpp, lnx_combine: every workgroup its 64 x 64
    block of the A operand, then a flag per block) instead of by attn_combine_kernel - three launches less per denoise step.  Same
    operations in the same order: the same bits, call after call on one handle.  (F16X2 has no one-launch GEMM + LayerNorm: its
    merge stays where it was, and the knob must change nothing.)"""
    z = np.load(os.path.join(GOLDEN, case))
    eng, _ = get_engine(int(z["ctx_dim"]), int(z["wseed"]), bool(z["joint"]))
    eng.set_step(int(z["step"]), "ddim")
    out = []
    try:
        for knob in (2, 0, 0, 0, 2):
            eng.set_tuning("small_cmb", knob)
            out.append(eng.denoise(z["x_T"][None], z["ctx"][None], precision=precision, want_pos=False)[0][0])
    finally:
        eng.set_tuning("small_cmb", 0)
    for o in out[1:]:
        np.testing.assert_array_equal(o, out[0])
    assert ade(out[1], z["vel"]) <= ADE_GATE


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", ["net_jmid_w256_a5k20t12_s50.npz", "net_imid_w32_a5k20t12_s50.npz", "ddpm_jmid_w32_a2k3t4_s10.npz"])
def test_output_kernel_with_fused_next_embedding_is_bit_identical(case, precision):
    """The output kernel of step i also embeds x for step i + 1 (one launch less per step): same bits as the separate
    embed_kernel, DDIM and DDPM."""
    z = np.load(os.path.join(GOLDEN, case))
    eng, _ = get_engine(int(z["ctx_dim"]), int(z["wseed"]), bool(z["joint"]))
    ddpm = case.startswith("ddpm")
    eng.set_step(int(z["step"]), "ddpm" if ddpm else "ddim")
    out = []
    try:
        for fuse in (1, 0):
            eng.set_tuning("fuse_embed", fuse)
            kw = {"z": z["z"][:, None]} if ddpm else {}
            out.append(eng.denoise(z["x_T"][None], z["ctx"][None], precision=precision, want_pos=False, **kw)[0][0])
    finally:
        eng.set_tuning("fuse_embed", 1)
        eng.set_step(int(z["step"]), "ddim")
    np.testing.assert_array_equal(out[0], out[1])


@pytest.mark.parametrize("precision", SPLIT_MODES)
def test_vt_through_lds_is_bit_identical_to_direct_stores(precision):
    """256x256 QKV kernel: V^T written in full rows through LDS (vt_staged_store) against the direct 8-byte stores and
    against the row-major V + transpose kernel, on a batch whose 64-token wave tiles straddle sequence boundaries
    (1200 = 18.75 x 64: those waves fall back to the direct path) and end in a partial row tile (E * 1200 % 256 != 0)."""
    eng, w = get_engine(256, 23, True)
    eng.set_step(4)
    E, A, K, T = 19, 5, 20, 12
    g = torch.Generator().manual_seed(13)
    ctx = torch.randn([E, A, 256], generator=g).cuda()
    x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
    out = {}
    try:
        for stage, novt in ((0, 0), (2, 0), (0, 1)):
            eng.set_tuning("vt_stage", stage)
            eng.set_tuning("no_vt_direct", novt)
            out[(stage, novt)] = eng.denoise(x_T, ctx, precision=precision, want_pos=False)[0].cpu().numpy()
    finally:
        eng.set_tuning("vt_stage", 0)
        eng.set_tuning("no_vt_direct", 0)
    np.testing.assert_array_equal(out[(0, 0)], out[(2, 0)])
    np.testing.assert_array_equal(out[(0, 0)], out[(0, 1)])
    with torch.no_grad():
        ref = O.denoise(w.tensors, ctx[:2].cpu(), x_T[:2].cpu(), sample=K, step=4, joint=True)
    assert ade(out[(0, 0)][:2], ref.numpy()) <= ADE_GATE


def test_linear1_tile_through_lds_is_bit_identical_to_the_elementwise_epilogue():
    """F16MX linear1 in the 64 x 128 wave tile (256 x 256 and 128 x 256 workgroup shapes): bias + ReLU + the fp16 plane written
    through LDS in whole lines (h1_staged_store, transposed product) against the element-wise epilogue ("h1_stage" = 2), on a
    batch that ends in a partial row tile (19 x 1200 % 256 != 0) - nn.TransformerEncoderLayer's linear1 + activation as built
    at MID/models/diffusion.py:161-166."""
    eng, w = get_engine(256, 23, True)
    eng.set_step(4)
    E, A, K, T = 19, 5, 20, 12
    g = torch.Generator().manual_seed(31)
    ctx = torch.randn([E, A, 256], generator=g).cuda()
    x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
    out = {}
    try:
        for variant in (0, 7):
            for stage in (0, 2):
                eng.set_tuning("gemm_h_variant", variant)
                eng.set_tuning("h1_stage", stage)
                out[(variant, stage)] = eng.denoise(x_T, ctx, precision="f16mx", want_pos=False)[0].cpu().numpy()
    finally:
        eng.set_tuning("h1_stage", 0)
        eng.set_tuning("gemm_h_variant", 0)
    for k in ((0, 2), (7, 0), (7, 2)):
        np.testing.assert_array_equal(out[k], out[(0, 0)])
    with torch.no_grad():
        ref = O.denoise(w.tensors, ctx[:2].cpu(), x_T[:2].cpu(), sample=K, step=4, joint=True)
    assert ade(out[(0, 0)][:2], ref.numpy()) <= ADE_GATE


KNOB_VALUES = [("gemm_h_variant", (1, 2, 3, 4, 5, 6, 7, 8)), ("h1_stage", (2,)), ("ln_fuse", (1, 2)), ("ln_rows", (64, 128)), ("attn_h_variant", (1, 2)),
               ("vt_stage", (1, 2, 3)), ("no_vt_direct", (1,)), ("attn_nsplit", (1, 3)), ("csl_swap", (2, 3)),
               ("out_traj", (1, 2)), ("attn_mx", (1, 2, 3)), ("fuse_embed", (0,)), ("attn_pack", (0,)), ("lanes", (1, 3)),
               ("bystander_lds", (100 * 1024,)), ("gemm_ng", (2,)), ("attn_pf", (2,)), ("attn_one_wg", (1,)), ("small_lnx", (2,)), ("cus", (128, 64)), ("attn_sm", (2,)), ("attn_prio", (1, 2)), ("small_cmb", (2,)), ("small_lnx2", (2,))]


@pytest.mark.parametrize("precision", SPLIT_MODES)
def test_every_tuning_knob_value_holds_parity(precision):
    """No combination of a mode with ONE knob off its default may leave the gate (a forced kernel variant that cannot
    produce an operand format the next kernel expects must be caught by the wiring, not by the user): 19 episodes
    (22 800 tokens: the large-tile kernels) and one scene (the small-M ones), 4 steps, against the oracle."""
    eng, w = get_engine(256, 23, True)
    eng.set_step(4)
    A, K, T = 5, 20, 12
    g = torch.Generator().manual_seed(29)
    ctx = torch.randn([19, A, 256], generator=g).cuda()
    x_T = torch.randn([19, K * A, T, 2], generator=g).cuda()
    with torch.no_grad():
        ref = O.denoise(w.tensors, ctx[:2].cpu(), x_T[:2].cpu(), sample=K, step=4, joint=True).numpy()
    base = eng.denoise(x_T, ctx, precision=precision, want_pos=False)[0].cpu().numpy()
    assert ade(base[:2], ref) <= ADE_GATE
    bad = []
    for knob, values in KNOB_VALUES:
        for v in values:
            try:
                eng.set_tuning(knob, v)
                big = eng.denoise(x_T, ctx, precision=precision, want_pos=False)[0].cpu().numpy()
                one = eng.denoise(x_T[:1], ctx[:1], precision=precision, want_pos=False)[0].cpu().numpy()
            finally:
                eng.set_tuning(knob, 2 if knob == "lanes" else 1 if knob in ("fuse_embed", "attn_pack") else 0)
            e_big, e_one = ade(big[:2], ref), ade(one, ref[:1])
            if not (e_big <= ADE_GATE and e_one <= ADE_GATE and ade(big, base) <= ADE_GATE):
                bad.append((knob, v, e_big, e_one, ade(big, base)))
    assert not bad, bad


def test_f16mx_attention_variants_hold_parity():
    """JMID_PREC_F16MX, head_dim 128: the default attention (bf8 logit corrections, one fp16 plane of P), the variant that keeps
    P_lo ("attn_mx" = 1) and F16X2's attention (= 2) differ at rounding level only: each within the gate of the oracle, and all
    three equally far from the exact-fp32 mode of the same library (the small-M 64 x 64 QKV tiles write the bf8 images with byte
    stores, the 19-episode batch through the staged 256 x 256 epilogue).  "attn_mx" = 3 (Q_lo as an fp16 plane, its bf8 image made
    in the attention kernel instead of by the QKV GEMM) must give the default's bits."""
    eng, w = get_engine(256, 23, True)
    A, K, T = 5, 20, 12
    try:
        for E, step in ((2, 50), (19, 4)):
            eng.set_step(step)
            g = torch.Generator().manual_seed(17 + E)
            ctx = torch.randn([E, A, 256], generator=g).cuda()
            x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
            out = {}
            for v in (0, 1, 2, 3):
                eng.set_tuning("attn_mx", v)
                out[v] = eng.denoise(x_T, ctx, precision="f16mx", want_pos=False)[0].cpu().numpy()
            np.testing.assert_array_equal(out[3], out[0])
            eng.set_tuning("attn_mx", 0)
            for var in (1, 3, 5):      # forced GEMM tile variants: F16X2's register-staged kernel (no bf8 images), two F16MX shapes
                eng.set_tuning("gemm_h_variant", var)
                o = eng.denoise(x_T, ctx, precision="f16mx", want_pos=False)[0].cpu().numpy()
                eng.set_tuning("gemm_h_variant", 0)
                if var == 1:
                    assert ade(o, out[2]) <= ADE_GATE      # F16X2's GEMMs and attention
                else:
                    np.testing.assert_array_equal(o, out[0])
            with torch.no_grad():
                ref = O.denoise(w.tensors, ctx[:2].cpu(), x_T[:2].cpu(), sample=K, step=step, joint=True).numpy()
            for v in (0, 1, 2):
                assert ade(out[v][:2], ref) <= ADE_GATE, (E, v)
            assert not np.array_equal(out[0], out[2])          # the knob really selects another kernel
            exact = eng.denoise(x_T, ctx, precision="f32", want_pos=False)[0].cpu().numpy()
            err = [ade(out[v], exact) for v in (0, 1, 2)]
            assert max(err) <= ADE_GATE and max(err) <= 1.25 * min(err), (E, err)     # no variant is the less accurate one
    finally:
        eng.set_tuning("attn_mx", 0)
        eng.set_tuning("gemm_h_variant", 0)
        eng.set_step(4)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_captured_denoise_loop_replays_bit_identically(precision):
    """Opt-in ("graph" = 1) for one-chunk calls: the 50-step loop runs eagerly the first time a shape is seen, is captured
    into a hipGraph the second time and replayed from then on - same kernels in the same order, so the same bits; a new
    step table, a tuning knob or another shape must never replay a stale graph."""
    eng, w = get_engine(256, 23, True)
    eng.set_step(10)
    g = torch.Generator().manual_seed(21)
    ctx = torch.randn([2, 5, 256], generator=g).numpy()
    x_T = torch.randn([2, 100, 12, 2], generator=g).numpy()
    p0 = torch.randn([2, 5, 2], generator=g).numpy()
    try:
        eng.set_tuning("graph", 2)
        ref = eng.denoise(x_T, ctx, p0, precision=precision)
        ref1 = eng.denoise(x_T[:1], ctx[:1], p0[:1], precision=precision)
        eng.set_tuning("graph", 1)
        n0 = eng.graph_replays()
        outs = [eng.denoise(x_T, ctx, p0, precision=precision) for _ in range(4)]        # eager, capture + launch, replay, replay
        assert eng.graph_replays() - n0 == 3
        for o in outs:
            np.testing.assert_array_equal(o[0], ref[0])
            np.testing.assert_array_equal(o[1], ref[1])
        # other inputs through the same graph (the graph holds workspace pointers, not the caller's data)
        o2 = eng.denoise(x_T[::-1].copy(), ctx[::-1].copy(), p0[::-1].copy(), precision=precision)
        np.testing.assert_array_equal(o2[0][::-1], ref[0])
        # another shape gets its own graph
        for _ in range(3):
            o1 = eng.denoise(x_T[:1], ctx[:1], p0[:1], precision=precision)
        np.testing.assert_array_equal(o1[0], ref1[0])
        # a new step table drops the captured loops
        eng.set_step(5)
        eng.set_tuning("graph", 2)
        ref5 = eng.denoise(x_T, ctx, p0, precision=precision)
        eng.set_tuning("graph", 1)
        for _ in range(3):
            o5 = eng.denoise(x_T, ctx, p0, precision=precision)
        np.testing.assert_array_equal(o5[0], ref5[0])
        assert not np.array_equal(ref5[0], ref[0])
        # device-mode call, same shape
        od = eng.denoise(torch.from_numpy(x_T).cuda(), torch.from_numpy(ctx).cuda(), torch.from_numpy(p0).cuda(),
                         precision=precision)
        np.testing.assert_array_equal(od[0].cpu().numpy(), ref5[0])
    finally:
        eng.set_tuning("graph", 0)
        eng.set_step(50)
