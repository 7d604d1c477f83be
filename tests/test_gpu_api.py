"""C-ABI behaviour on the GPU: sweep metrics kernel, device-pointer encode, error codes, handle lifetime."""
import ctypes as C
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from safe_interactive_crowdnav_amd import _lib
from safe_interactive_crowdnav_amd.engine import JmidEngine, JmidError
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims


@pytest.fixture(scope="module")
def eng():
    e = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=32), 3), joint=True, step=2)
    yield e
    e.close()


def test_episode_metrics_match_numpy(eng):
    rng = np.random.default_rng(0)
    E, K, A, T = 7, 20, 5, 12
    pos = rng.standard_normal((E, K, A, T, 2)).astype(np.float32)
    gt = rng.standard_normal((E, A, T, 2)).astype(np.float32)
    out = eng.episode_metrics(pos, gt)
    d = np.linalg.norm(pos.astype(np.float64) - gt[:, None].astype(np.float64), axis=-1)     # [E,K,A,T]
    ade = d.mean(axis=(2, 3))            # per sample, agent-and-time mean (evaluation.py:20-26 per element)
    fde = d[..., -1].mean(axis=2)
    ref = np.stack([ade.mean(1), ade.min(1), fde.mean(1), fde.min(1)], axis=1)
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-6)
    out_d = eng.episode_metrics(torch.from_numpy(pos).cuda(), torch.from_numpy(gt).cuda())
    eng.synchronize()
    np.testing.assert_array_equal(out_d.cpu().numpy(), out)


def test_encode_device_pointers_match_host(eng):
    g = torch.Generator().manual_seed(2)
    x_st = torch.randn([11, 6, 6], generator=g)
    nbr = torch.randn([11, 2, 6, 6], generator=g)
    em = torch.rand([11, 2], generator=g)
    a = eng.encode(x_st.numpy(), nbr.numpy(), em.numpy())
    b = eng.encode(x_st.cuda(), nbr.cuda(), em.cuda())
    eng.synchronize()
    np.testing.assert_array_equal(b.cpu().numpy(), a)


def test_error_codes(eng):
    ctx = np.zeros((1, 2, 32), np.float32)
    with pytest.raises(JmidError) as ei:       # T beyond the positional-encoding table (max_len = 24)
        eng.denoise(np.zeros((1, 4, 25, 2), np.float32), ctx, want_pos=False)
    assert ei.value.code == -1 and "max_len" in str(ei.value)
    with pytest.raises(JmidError) as ei:
        eng.net_eval(np.zeros((1, 4, 4, 2), np.float32), ctx, step_idx=99)
    assert ei.value.code == -1
    with pytest.raises(ValueError):
        eng.denoise(np.zeros((1, 5, 4, 2), np.float32), ctx)          # 5 rows is not a multiple of A = 2
    with pytest.raises(JmidError):
        eng.set_tuning("no_such_knob", 1)
    lib = _lib.load_library()
    h = _lib.Handle()
    assert lib.jmid_create(C.byref(h), 0, 7, 32, 3, 4, 6) == -1       # bad net_kind
    assert lib.jmid_create(C.byref(h), 0, 1, 48, 3, 4, 6) == -1       # ctx_dim not a multiple of 32
    assert lib.jmid_create(C.byref(h), 99, 1, 32, 3, 4, 6) == -1      # no such device


def test_unfinalized_engine_refuses_to_compute():
    lib = _lib.load_library()
    h = _lib.Handle()
    assert lib.jmid_create(C.byref(h), 0, 1, 32, 3, 4, 6) == 0
    x = np.zeros((1, 2, 4, 2), np.float32)
    ctx = np.zeros((1, 1, 32), np.float32)
    out = np.zeros_like(x)
    rc = lib.jmid_net_eval(h, 1, 1, 2, 4, 0, x.ctypes.data_as(C.c_void_p), ctx.ctypes.data_as(C.c_void_p), 0,
                           out.ctypes.data_as(C.c_void_p), 0)
    assert rc == -2 and b"finalize" in lib.jmid_last_error(h)
    w = np.zeros(5, np.float32)
    assert lib.jmid_load_weight(h, b"concat1._layer.bias", w.ctypes.data_as(C.c_void_p), 5) == -1    # wrong size
    assert lib.jmid_load_weight(h, b"nonsense", w.ctypes.data_as(C.c_void_p), 5) == -1
    assert lib.jmid_finalize_weights(h) == -2                                                        # weights missing
    assert lib.jmid_destroy(h) == 0


def test_two_engines_coexist():
    """Two handles (e.g. iMID and JMID predictors of two policies) keep separate weights and streams."""
    w1 = JMIDWeights.from_seed(NetDims(ctx_dim=32), 1)
    w2 = JMIDWeights.from_seed(NetDims(ctx_dim=32), 2)
    e1, e2 = JmidEngine(w1, joint=True, step=2), JmidEngine(w2, joint=False, step=2)
    g = torch.Generator().manual_seed(0)
    ctx = torch.randn([1, 2, 32], generator=g).numpy()
    x = torch.randn([1, 6, 4, 2], generator=g).numpy()
    a1, _ = e1.denoise(x, ctx, want_pos=False)
    b, _ = e2.denoise(x, ctx, want_pos=False)
    a2, _ = e1.denoise(x, ctx, want_pos=False)
    np.testing.assert_array_equal(a1, a2)
    assert np.abs(a1 - b).max() > 1e-3
    e1.close()
    e2.close()


@pytest.mark.parametrize("which", ["side", "default"])
def test_device_mode_calls_are_ordered_against_the_callers_stream(which):
    """JMID_MEM_DEVICE buffers are produced and consumed on the CALLER's stream (include/jmid_hip.h, Conventions): the
    library's private stream waits for what the caller enqueued before the call and the caller's stream waits for the
    call's last kernel.  Inputs that are still being computed on a non-default torch stream when the call is made, and
    outputs consumed on that stream right after it, must give the synchronous answer.  "default": the same with torch's DEFAULT
    stream (the legacy null stream) as the busy producer - the handle's streams are non-blocking streams since round 6, so nothing but
    the library's own events orders them against it."""
    import torch
    from safe_interactive_crowdnav_amd.engine import JmidEngine
    from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
    w = JMIDWeights.from_seed(NetDims(ctx_dim=32), 3)
    eng = JmidEngine(w, joint=True, step=10)
    E, A, K, T = 6, 3, 4, 6
    g = torch.Generator().manual_seed(2)
    ctx0 = torch.randn([E, A, 32], generator=g).cuda()
    x0 = torch.randn([E, K * A, T, 2], generator=g).cuda()
    p0 = torch.randn([E, A, 2], generator=g).cuda()
    big = torch.randn([4096, 4096], device="cuda")
    torch.cuda.synchronize()
    ref_vel, ref_pos = eng.denoise(x0 * 2.0 - x0, ctx0 + 0.0, p0, precision="f32")     # default stream, synchronous reference
    eng.synchronize()
    side = torch.cuda.Stream() if which == "side" else torch.cuda.default_stream()
    for _ in range(3):
        with torch.cuda.stream(side):
            for _ in range(8):                      # keep the side stream busy: the inputs below are produced behind this
                big = (big @ big).clamp_(-1.0, 1.0)
            x_in = x0 * 2.0 - x0                    # producer kernels of the inputs, on the side stream
            c_in = ctx0 + 0.0
            vel, pos = eng.denoise(x_in, c_in, p0, precision="f32")    # f32: returns without a host sync
            chk = (pos - ref_pos).abs().max() + (vel - ref_vel).abs().max()     # consumer on the side stream
            worst = float(chk.item())               # read back on the side stream too: torch's streams are non-blocking,
        assert worst == 0.0                         # the default stream would not wait for the consumer
    eng.close()


def test_tuning_knobs_belong_to_their_handle():
    """jmid_set_tuning acts on the handle it is called on, never on another engine of the process."""
    import numpy as np
    import torch
    from safe_interactive_crowdnav_amd.engine import JmidEngine
    from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
    w = JMIDWeights.from_seed(NetDims(ctx_dim=256), 3)
    a, b = JmidEngine(w, joint=True, step=4), JmidEngine(w, joint=True, step=4)
    g = torch.Generator().manual_seed(4)
    ctx = torch.randn([1, 5, 256], generator=g).numpy()
    x_T = torch.randn([1, 40, 12, 2], generator=g).numpy()
    base = b.denoise(x_T, ctx, precision="f16x3", want_pos=False)[0]
    a.set_tuning("attn_nsplit", 1)                  # changes the order a sequence's keys are summed in: rounding-level
    a.set_tuning("lanes", 3)
    a.set_chunk_episodes(1)
    out_a = a.denoise(x_T, ctx, precision="f16x3", want_pos=False)[0]
    out_b = b.denoise(x_T, ctx, precision="f16x3", want_pos=False)[0]
    np.testing.assert_array_equal(out_b, base)      # engine b never saw engine a's knobs
    assert not np.array_equal(out_a, base) and np.abs(out_a - base).max() < 1e-4
    for key in ("attn_abl", "gemm_abl"):            # timing ablations (wrong results) do not exist in the production build
        with pytest.raises(Exception):
            a.set_tuning(key, 1)
    a.close()
    b.close()


def test_two_handles_run_concurrently_from_two_host_threads():
    """Several handles may share a GPU and be driven from separate host threads (an iMID and a JMID predictor, two
    policies).  ctypes drops the GIL for the duration of a call, so the two calls below really overlap: different nets, different
    modes, different knobs - each thread gets, every time, the bits its engine produces alone.  (Exercises the per-device
    once-only function attributes under their mutex, the thread-local tuning pointer and the per-handle streams / workspaces.)"""
    import threading
    import numpy as np
    import torch
    from safe_interactive_crowdnav_amd.engine import JmidEngine
    from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
    g = torch.Generator().manual_seed(8)
    jobs = []
    for joint, prec, E, knob in ((True, "f16mx", 3, ("lanes", 1)), (False, "f16x3", 5, ("lanes", 3))):
        eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 11 + int(joint)), joint=joint, step=5)
        eng.set_tuning(*knob)
        ctx = torch.randn([E, 5, 256], generator=g).numpy()
        x_T = torch.randn([E, 100, 12, 2], generator=g).numpy()
        p0 = torch.randn([E, 5, 2], generator=g).numpy()
        alone = [np.array(t) for t in eng.denoise(x_T, ctx, p0, precision=prec)]
        jobs.append((eng, ctx, x_T, p0, prec, alone))
    errors, start = [], threading.Barrier(2)

    def work(eng, ctx, x_T, p0, prec, alone):
        try:
            start.wait()
            for _ in range(12):
                vel, pos = eng.denoise(x_T, ctx, p0, precision=prec)
                np.testing.assert_array_equal(np.asarray(vel), alone[0])
                np.testing.assert_array_equal(np.asarray(pos), alone[1])
        except Exception as e:          # noqa: BLE001 - reported by the main thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for eng, *_ in jobs:
        eng.close()
    assert not errors, errors


@pytest.mark.parametrize("ctx_dim,precision", [(32, "f32"), (256, "f16x3"), (256, "f16mx")])
@pytest.mark.parametrize("E,A,K,T,k", [(1, 3, 100, 8, 15), (1, 5, 20, 12, 20), (3, 2, 12, 6, 4)])
def test_predict_entry_equals_the_staged_calls(ctx_dim, precision, E, A, K, T, k):
    """jmid_predict (what predict_ret_best() issues per MPC step: mid_sim_wrapper.py:482-510) = jmid_encode -> jmid_denoise ->
    jmid_topk chained on the stream: the same bits as the three calls, for the shipped shape, for k == K (no ranking) and for
    several episodes; repeated calls (the pinned staging buffers are reused) agree."""
    e = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=ctx_dim), 3), joint=True, step=2)
    try:
        g = torch.Generator().manual_seed(E * 100 + A * 10 + k)
        x_st = torch.randn([E * A, 6, 6], generator=g).numpy()
        nbr = torch.randn([E * A, 2, 6, 6], generator=g).numpy()
        em = torch.rand([E * A, 2], generator=g).numpy()
        x_T = torch.randn([E, K * A, T, 2], generator=g).numpy()
        p0 = torch.randn([E, A, 2], generator=g).numpy()
        ctx = e.encode(x_st, nbr, em).reshape(E, A, -1)
        _, pos = e.denoise(x_T, ctx, p0, dt=0.25, precision=precision, want_vel=False)
        for _ in range(2):
            out, lw = e.predict(x_st, nbr, em, x_T, p0, k, dt=0.25, precision=precision)
            if k < K:
                sel_ref, lw_ref = e.topk(pos, k)
                np.testing.assert_array_equal(out, sel_ref)
                np.testing.assert_array_equal(lw, lw_ref)
            else:
                assert lw is None
                np.testing.assert_array_equal(out, pos)
        with pytest.raises(JmidError) as ei:
            e.predict(x_st, nbr, em, x_T, p0, K + 1, precision=precision)
        assert ei.value.code == -1
    finally:
        e.close()


def _one_scene(seed=5):
    g = torch.Generator().manual_seed(seed)
    A, K, T = 5, 20, 12
    return (torch.randn([1, K * A, T, 2], generator=g).numpy(), torch.randn([1, A, 256], generator=g).numpy(),
            torch.randn([1, A, 2], generator=g).numpy())


@pytest.mark.expects_timeout
def test_a_workgroup_that_gives_up_waiting_comes_back_as_timeout_and_the_retry_is_correct():
    """The one-launch GEMM + LayerNorm of one-scene F16MX calls (gemm_small.hpp, OUT_LNX) waits for partner workgroups with BOUNDED polls.
    Provoked here (diagnostics knobs: one workgroup never publishes its statistics, poll budget 2 000): the library returns
    JMID_ETIMEOUT - not JMID_ERANGE: nothing left the fp16 range, no exact-fp32 rerun - counts it (jmid_timeout_count), drops the kernel
    for this handle, and the engine's single retry in the SAME precision returns the bits of the unfused pair; later calls stay there."""
    from safe_interactive_crowdnav_amd import engine as EN
    x_T, ctx, p0 = _one_scene()
    eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 4), joint=True, step=10)
    try:
        eng.set_tuning("small_lnx", 2)
        ref = eng.denoise(x_T, ctx, p0, precision="f16mx")                    # GEMM + add_ln2
        eng.set_tuning("small_lnx", 0)
        fused = eng.denoise(x_T, ctx, p0, precision="f16mx")                  # the one-launch form: same bits, nobody gives up
        for a, b in zip(fused, ref):
            np.testing.assert_array_equal(a, b)
        assert eng.timeout_count() == 0
        n0 = len(EN.TIMEOUT_EVENTS)
        eng.set_tuning("lnx_withhold", 1)
        eng.set_tuning("lnx_polls", 2000)
        t0 = time.perf_counter()
        got = eng.denoise(x_T, ctx, p0, precision="f16mx")                    # first attempt times out, the retry runs the pair
        dt = time.perf_counter() - t0
        assert len(EN.TIMEOUT_EVENTS) == n0 + 1 and "gave up waiting" in EN.TIMEOUT_EVENTS[-1][1]
        assert eng.timeout_count() == 1 and eng.erange_count() == 0
        for a, b in zip(got, ref):
            np.testing.assert_array_equal(a, b)
        assert dt < 5.0, dt                                                    # the launches behind the first timeout leave early
        again = eng.denoise(x_T, ctx, p0, precision="f16mx")                  # the handle stays on the pair: no second timeout
        assert eng.timeout_count() == 1 and len(EN.TIMEOUT_EVENTS) == n0 + 1
        for a, b in zip(again, ref):
            np.testing.assert_array_equal(a, b)
    finally:
        eng.close()


@pytest.mark.expects_timeout
def test_timeout_is_reported_to_a_c_caller_with_its_own_status_code():
    """Straight through the C ABI (no engine retry): JMID_ETIMEOUT = -6 with a message, the next call on the handle JMID_OK."""
    import ctypes as C
    from safe_interactive_crowdnav_amd import _lib
    x_T, ctx, p0 = _one_scene(6)
    eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 4), joint=True, step=4)
    try:
        eng.set_tuning("lnx_withhold", 1)
        eng.set_tuning("lnx_polls", 500)
        vel = np.empty_like(x_T)
        args = (eng._h, 1, 5, 20, 12, C.c_void_p(x_T.ctypes.data), C.c_void_p(ctx.ctypes.data), None, C.c_float(0.25),
                _lib.PRECISIONS["f16mx"], C.c_void_p(vel.ctypes.data), None, _lib.MEM_HOST)
        rc = eng._lib.jmid_denoise(*args)
        assert rc == -6 and _lib.ERR_NAMES[rc] == "JMID_ETIMEOUT"
        assert b"same precision" in eng._lib.jmid_last_error(eng._h)
        assert eng._lib.jmid_timeout_count(eng._h) == 1 and eng._lib.jmid_erange_count(eng._h) == 0
        assert eng._lib.jmid_denoise(*args) == 0 and np.isfinite(vel).all()
    finally:
        eng.close()


@pytest.mark.expects_timeout
def test_two_handles_each_on_one_scene_in_f16mx_from_two_threads():
    """Two handles, BOTH on one cfg2 scene in F16MX, from two host threads at once: each launch of the one-launch GEMM + LayerNorm then
    shares the chip with the other handle's, which is exactly when "all partner workgroups are resident" is not given by construction.
    Either everything comes back with the bits of each handle alone, or a handle reports a timeout, retries and still returns those
    bits - never garbage, never a hang."""
    import threading
    x_T, ctx, p0 = _one_scene(7)
    engs = [JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 4), joint=True, step=50) for _ in range(2)]
    try:
        alone = [e.denoise(x_T, ctx, p0, precision="f16mx") for e in engs]
        for a, b in zip(alone[0], alone[1]):
            np.testing.assert_array_equal(a, b)
        out = [[None] * 6 for _ in engs]
        def run(i):
            for r in range(6):
                out[i][r] = engs[i].denoise(x_T, ctx, p0, precision="f16mx")
        th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        [t.start() for t in th]
        [t.join(timeout=120) for t in th]
        assert not any(t.is_alive() for t in th)
        for i in range(2):
            for r in range(6):
                for a, b in zip(out[i][r], alone[0]):
                    np.testing.assert_array_equal(a, b)
            assert engs[i].erange_count() == 0 and engs[i].timeout_count() in (0, 1)
        print("timeouts per handle:", [e.timeout_count() for e in engs])
    finally:
        for e in engs:
            e.close()
