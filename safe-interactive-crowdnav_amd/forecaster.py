"""Drop-in predictor: same call surface as the reference's ``HumanTrajectoryForecasterSim``.

Reference: ``sicnav_diffusion/JMID/mid_sim_wrapper.py:173-510`` (class at ``:207``).  The only caller,
``SICNavAcados`` (``sicnav_diffusion/policy/sicnav_acados.py:996-1000, 1171-1182, 1641-1651``), uses

    f = HumanTrajectoryForecasterSim(env_config, mid_config_file)
    f.num_hist_frames
    f.update_state_hists(robot_state, human_states, time_stamp)
    forecasts, log_weights = f.predict_ret_best()      # [N, k, H+1, 2] f64, [N, k] f64

and nothing else.  Here the history buffers and the scene/batch construction are NumPy on the host
(``scene.py``), and everything from the context encoder to the integrated sample trajectories runs in the HIP
library through the C ABI (``engine.py`` -> ``include/jmid_hip.h``).  There is no CPU fallback for that part.

Arithmetic (``precision``): since round 6 the class default is the mode ``bench.py`` quotes, ``"f16mx"`` (fp16 activation x
split-fp16 weight, ``A_hi . W_hi`` on the fp16 matrix cores plus the correction term as ONE bf8 x bf8 MFMA per 64-deep block;
5e-6 m on the 50-step cfg3 sample, worst fixture 5.7e-5 m against the 1e-4 m gate) TOGETHER WITH ``self_check=True``: every
accuracy figure in this repository is on seeded random-init weights (the trained blobs are absent from the reference), so the
first call of every input shape is also run in ``"f16x3"`` - fp32-class three-term split-fp16 products, ~1e-6 m from the
reference (the fp32-vs-fp64 noise floor) - on the call's own inputs, and the instance moves to ``"f16x3"`` for good (with a
``RuntimeWarning``) when the mean displacement between the two exceeds ``self_check_tol`` (5e-5 m: half the gate, so a mode that
passes holds the gate with 2x margin on that checkpoint and shape).  ``"f16x2"`` (the same products with fp16 correction terms),
``"f16x3"`` and ``"f32"`` (exact-fp32 MFMA) can be asked for by name; ``self_check`` only acts on ``"f16mx"`` / ``"f16x2"``.  If an activation ever leaves the fp16 range (``JMID_ERANGE``) the call is repeated in the exact-fp32 mode: same
result, ~5x the latency - counted in ``erange_fallbacks`` / ``forecaster.ERANGE_FALLBACKS`` and warned about once.

RNG contract (``rng_compat``): ``x_T`` is always the first draw of torch's CPU default generator
(``MID/models/diffusion.py:499``).  The reference also draws one (unused, DDIM) ``randn_like(x_T)`` per reverse step
(``:509``) on the device ``x_T`` was moved to: the CPU generator in a CPU-only run (what the golden captures are),
the CUDA generator on a GPU host (``MID/mid.py:91``).  ``rng_compat="cpu"`` consumes those draws from the CPU
generator, ``"cuda"`` from the device generator (the CPU generator then advances by ``x_T`` only; the device generator is
moved by what the draws would consume without launching them, ``advance_cuda_generator``), and the default
``"auto"`` does what the reference itself would do on this host (``"cuda"`` iff ``torch.cuda.is_available()``).

Differences from the reference that a caller can observe:
  * the engine (weights on the GPU) is cached across instances: the reference rebuilds the model and reloads the
    checkpoint at the start of every episode (``sicnav_acados.py:1171``); the cache key holds the checkpoint file's
    identity (path, size, mtime) or the weight checksum (computed once per ``JMIDWeights`` object), the engine is
    one per (weights, net, device, history length, step size) and calls on it are serialised by a lock;
  * ``model_path`` may point to the neutral ``.npz`` written by ``export_checkpoint`` instead of the pickled
    ``nn.ModuleDict`` container (which needs the reference tree on ``sys.path`` to unpickle);
  * a too-short history raises ``HistoryTooShortError`` (a ``TypeError``, like the reference's failure mode).
"""
from __future__ import annotations

import configparser
import os
import time
import warnings
from threading import Lock, RLock
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import yaml

from . import scene as SC
from .engine import JmidEngine, JmidError
from .kde import most_likely_samples
from .weights import JMIDWeights, NetDims

_ENGINE_CACHE: Dict[tuple, JmidEngine] = {}
_ENGINE_LOCKS: Dict[int, RLock] = {}
_WEIGHTS_CACHE: Dict[tuple, JMIDWeights] = {}     # (path, size, mtime_ns, dims) -> weights (re-created every episode)
_CACHE_LOCK = Lock()            # guards the three caches above (forecasters may be built from several threads)
ERANGE_FALLBACKS = 0            # calls of this process that were repeated in "f32" after JMID_ERANGE
SELF_CHECK_DOWNGRADES = 0       # instances whose opt-in precision was replaced by "f16x3" by the first-call self check
# what HumanTrajectoryForecasterSim's keyword arguments default to (safe_interactive_crowdnav_amd.install(**defaults) edits it)
DEFAULTS = {"device_id": 0, "precision": "f16mx", "rng_compat": "auto", "self_check": True, "device_topk": True}


_PHILOX_STEP: Dict[Tuple[int, Tuple[int, ...]], int] = {}     # (device, shape) -> what one randn_like of that shape adds to the Philox offset


def advance_cuda_generator(device_id: int, shape: Tuple[int, ...], n: int) -> None:
    """Leave device ``device_id``'s default generator where ``n`` successive ``torch.randn_like(z)`` (z fp32 of ``shape`` on that
    device) leave it - the reference's unused per-step draws (``MID/models/diffusion.py:509``) - without launching them: a CUDA
    generator is (seed, Philox offset), and a draw of a given shape adds a fixed amount to the offset.  The first call for a
    (device, shape) makes ONE real draw to learn that amount; 49 launches per cfg2 call (0.2 ms of host time) otherwise."""
    if n <= 0:
        return
    gen = torch.cuda.default_generators[device_id] if len(torch.cuda.default_generators) > device_id else None
    if gen is None:                               # (CUDA not initialised yet: the first tensor on the device does it)
        torch.empty(1, device=f"cuda:{device_id}")
        gen = torch.cuda.default_generators[device_id]
    key = (int(device_id), tuple(int(v) for v in shape))
    if not (hasattr(gen, "get_offset") and hasattr(gen, "set_offset")):
        # an older torch (the reference pins 1.13.1) has no Philox-offset accessors: make the draws for real, as the reference does
        for _ in range(n):
            torch.randn(key[1], device=f"cuda:{device_id}")
        return
    step = _PHILOX_STEP.get(key)
    if step is None:
        before = gen.get_offset()
        torch.randn(key[1], device=f"cuda:{device_id}")
        step = gen.get_offset() - before
        _PHILOX_STEP[key] = step
        n -= 1
    if n > 0:
        gen.set_offset(gen.get_offset() + n * step)


def load_weights(model_path: str, dims: NetDims) -> JMIDWeights:
    """Weights from ``model_path``: the neutral ``.npz`` (preferred; also looked up next to a ``.pt`` path), or the
    reference's two-part checkpoint ``{"encoder": nn.ModuleDict, "ddpm": state_dict}`` (``MID/mid.py:1231-1232,
    1291, 1501-1509``) when it can be unpickled in this interpreter."""
    cand = [model_path] if model_path.endswith(".npz") else [os.path.splitext(model_path)[0] + ".npz", model_path]
    for p in cand:
        if not os.path.exists(p):
            continue
        st = os.stat(p)
        key = (os.path.abspath(p), st.st_size, st.st_mtime_ns, dims)
        with _CACHE_LOCK:
            if key in _WEIGHTS_CACHE:
                return _WEIGHTS_CACHE[key]
        if p.endswith(".npz"):
            w = JMIDWeights.load(p)
            if w.dims != dims:
                raise ValueError(f"{p} holds dims {w.dims}, config asks for {dims}")
        else:
            ckpt = torch.load(p, map_location="cpu", weights_only=False)
            w = JMIDWeights.from_reference_state(dims, ckpt["ddpm"], ckpt["encoder"])
        with _CACHE_LOCK:
            w = _WEIGHTS_CACHE.setdefault(key, w)
        return w
    raise FileNotFoundError(f"no checkpoint at {model_path} (or its .npz export)")


class ForecasterSimSuper:
    """History bookkeeping, identical to ``mid_sim_wrapper.py:172-204``."""

    def init_super(self, env_config):
        self.prev_states_lock = Lock()
        if env_config is None:     # mid_sim_wrapper.py:175-178: a file relative to the CWD (NoSectionError below when it is absent)
            env_config = configparser.RawConfigParser()
            env_config.read("./src/human_traj_forecaster/configs/env_utias_vicon.config")
        self.publish_freq = env_config.getfloat("human_trajectory_forecaster", "publish_freq")
        self.time_step = env_config.getfloat("env", "time_step")
        assert (self.time_step * 100).is_integer(), \
            "please only specify human time step to a hundredth of a second"
        self.num_hist_frames = env_config.getint("human_trajectory_forecaster", "past_num_frames")
        self.predict_horizon = env_config.getint("human_trajectory_forecaster", "prediction_horizon")
        self.num_ret_samples = env_config.getint("human_trajectory_forecaster", "num_samples")
        self.num_hums = env_config.getint("sim", "human_num")
        self.prev_states = [[] for _ in range(self.num_hums)]
        self.prev_robot_states = []

    def update_state_hists(self, robot_state, human_states, time_stamp):
        for i in range(self.num_hums):
            self.prev_states[i].append([*human_states[i].position, time_stamp])
            if len(self.prev_states[i]) > self.num_hist_frames:
                self.prev_states[i].pop(0)
        self.prev_robot_states.append([*robot_state.position, time_stamp])


class _ModelInfo:
    """What the reference exposes as ``.model`` / ``.mid_model`` (an ``MID`` object): kept as a plain record."""

    def __init__(self, config, engine, num_samples):
        self.config = config
        self.engine = engine
        self.num_samples = num_samples


class HumanTrajectoryForecasterSim(ForecasterSimSuper):
    def __init__(self, env_config=None, mid_config_file=None, *, weights: Optional[JMIDWeights] = None,
                 device_id: Optional[int] = None, precision: Optional[str] = None, rng_compat: Optional[str] = None,
                 self_check: Optional[bool] = None, self_check_tol: float = 5e-5, device_topk: Optional[bool] = None,
                 lib_path: Optional[str] = None):
        # (lib_path: another build of the library - tests run both flavours in one process; the product leaves it at None)
        # keyword arguments left at None take the process-wide defaults (``DEFAULTS``; ``install(**defaults)`` sets them for
        # a caller that constructs the class with the reference's two positional arguments only, sicnav_acados.py:998-1000)
        device_id = DEFAULTS["device_id"] if device_id is None else device_id
        precision = DEFAULTS["precision"] if precision is None else precision
        rng_compat = DEFAULTS["rng_compat"] if rng_compat is None else rng_compat
        self_check = DEFAULTS["self_check"] if self_check is None else self_check
        device_topk = DEFAULTS["device_topk"] if device_topk is None else device_topk
        self.init_super(env_config)
        self.precision = precision
        self.self_check = bool(self_check) and precision in ("f16mx", "f16x2")
        self.self_check_tol = float(self_check_tol)
        self._checked_shapes = set()
        self.device_topk = bool(device_topk)
        self.erange_fallbacks = 0
        self.timings: Dict[str, float] = {}     # ms of the last predict_ret_best(): scene, device, topk, assemble
        if rng_compat not in ("auto", "cpu", "cuda"):
            raise ValueError("rng_compat must be 'auto', 'cpu' or 'cuda'")
        self.rng_compat = rng_compat if rng_compat != "auto" else ("cuda" if torch.cuda.is_available() else "cpu")
        self._init_MID(mid_config_file, weights, device_id, lib_path)

    def _init_MID(self, mid_config_file, weights, device_id, lib_path=None):
        with open(mid_config_file) as f:
            cfg = yaml.safe_load(f)
        self.config = cfg
        dims = NetDims(ctx_dim=int(cfg["encoder_dim"]), tf_layer=int(cfg["tf_layer"]))
        self.joint = cfg["diffnet"] == "JointPredictionTransformerConcatLinear"
        if not self.joint and cfg["diffnet"] != "TransformerConcatLinear":
            raise ValueError(f"unsupported diffnet {cfg['diffnet']!r}")
        self.num_samples = int(cfg["num_samples"])
        self.step_size = int(cfg["step_size"])
        # the sampler runs with the YAML horizon / history length (MID/mid.py:1260-1261), the wrapper allocates its
        # output with the env-config ones (mid_sim_wrapper.py:494): they have to agree
        if int(cfg["prediction_horizon"]) != self.predict_horizon:
            raise ValueError("prediction_horizon differs between the MID yaml and [human_trajectory_forecaster]")
        if int(cfg["maximum_history_length"]) != self.num_hist_frames - 1:
            raise ValueError("maximum_history_length must equal past_num_frames - 1")
        if weights is None:
            weights = load_weights(cfg["model_path"], dims)
        key = (weights.checksum(), self.joint, device_id, self.num_hist_frames, self.step_size, lib_path)
        with _CACHE_LOCK:
            eng = _ENGINE_CACHE.get(key)
            if eng is None:
                eng = JmidEngine(weights, joint=self.joint, device_id=device_id, hist_len=self.num_hist_frames,
                                 step=self.step_size, lib_path=lib_path)
                _ENGINE_CACHE[key] = eng
                _ENGINE_LOCKS[id(eng)] = RLock()
            self.engine = eng
            self._engine_lock = _ENGINE_LOCKS[id(eng)]
        self.mid_model = _ModelInfo(cfg, eng, self.num_samples)
        self.model = self.mid_model

    # ------------------------------------------------------------------------------------------ prediction
    def _snapshot(self):
        with self.prev_states_lock:           # mid_sim_wrapper.py:251-258
            return [list(map(list, h)) for h in self.prev_states], list(map(list, self.prev_robot_states))

    def _denoise(self, x_T, ctx, p0, want_pos=True):
        """One denoise call in ``self.precision``; JMID_ERANGE -> the same call in exact fp32 (counted, warned about once)."""
        pos, fell_back = denoise_with_fallback(self.engine, x_T, ctx, p0, self.time_step, self.precision, want_pos)
        if fell_back:
            if not self.erange_fallbacks:
                warnings.warn("JMID_ERANGE: an activation left the fp16 range; the call was repeated in the exact-fp32 mode "
                              "(same result, ~5x the latency).  Construct the forecaster with precision='f32' if this "
                              "checkpoint does it often.", RuntimeWarning, stacklevel=3)
            self.erange_fallbacks += 1
        return pos

    def _self_check(self, x_T, ctx, p0, pos):
        """First call per shape in an opt-in mode: the same inputs once more in "f16x3"; downgrade when they disagree."""
        global SELF_CHECK_DOWNGRADES
        key = tuple(x_T.shape)
        if key in self._checked_shapes:
            return pos
        self._checked_shapes.add(key)
        ref, _ = denoise_with_fallback(self.engine, x_T, ctx, p0, self.time_step, "f16x3")
        delta = float(np.linalg.norm(pos - ref, axis=-1).mean())
        self.self_check_delta = delta
        if delta > self.self_check_tol:
            warnings.warn(f"precision={self.precision!r} differs from 'f16x3' by {delta:.2e} m mean displacement on this "
                          f"checkpoint (> {self.self_check_tol:.1e}): this forecaster now runs 'f16x3'", RuntimeWarning,
                          stacklevel=3)
            self.precision, self.self_check = "f16x3", False
            SELF_CHECK_DOWNGRADES += 1
            return ref
        return pos

    def get_most_likely_samples(self, forecasts):
        """mid_sim_wrapper.py:440-441: the ``num_ret_samples`` most likely of the sampled joint futures under the per-step
        Gaussian KDE (``get_most_likely_samples``, ``:14-169``, joint branch).  forecasts [K, A, H, 2] (tensor or array) ->
        (kept [A, k, H, 2], log-weights [A, k]) as float32 torch tensors, ascending likelihood like the reference's
        ``argsort(...)[-k:]``.  On the device (``jmid_topk``) while the shape fits it, on the host twin otherwise."""
        f = forecasts.detach().cpu().numpy() if isinstance(forecasts, torch.Tensor) else np.asarray(forecasts)
        f = np.ascontiguousarray(f, dtype=np.float32)
        K, A, H, _ = f.shape
        k = self.num_ret_samples
        if self.device_topk and k <= K and topk_fits_device(A, K, H):
            with self._engine_lock:
                sel, lw = self.engine.topk(f[None], k)
            sel, lw = sel[0], lw[0]
        else:
            sel, lw = most_likely_samples(f, k)
        return torch.from_numpy(np.ascontiguousarray(sel)), torch.from_numpy(np.ascontiguousarray(lw))

    def predict_ret_best(self) -> Tuple[np.ndarray, np.ndarray]:
        """mid_sim_wrapper.py:482-510 -> (forecasts [N, k, H+1, 2] float64, log-weights [N, k] float64)."""
        t0 = time.perf_counter()
        prev, rob = self._snapshot()
        hum_xy, rob_xy, pose_now = SC.frame_table(prev, rob, self.time_step, self.num_hist_frames)
        sb = SC.build_scene(hum_xy, rob_xy, self.time_step, self.predict_horizon, self.num_hist_frames)
        A, K, H, k = len(sb.ids_in), self.num_samples, self.predict_horizon, self.num_ret_samples
        # RNG contract (module docstring): x_T is the first draw of the CPU default generator; the per-step z of the
        # reference (unused by DDIM) comes from the generator of the device the reference would run on
        x_T = torch.randn([K * A, H, 2])
        stride = int(100 / self.step_size)
        n_z = sum(1 for t in range(100, 0, -stride) if t > 1)
        if self.rng_compat == "cpu":
            for _ in range(n_z):
                torch.randn_like(x_T)
        else:
            advance_cuda_generator(self.engine.device_id, tuple(x_T.shape), n_z)
        t1 = time.perf_counter()
        with self._engine_lock:               # the engine is shared between forecaster instances and not re-entrant
            if self.engine.step != self.step_size or self.engine.sampling != "ddim":
                self.engine.set_step(self.step_size, "ddim")   # eval_sicnav hard-codes sampling="ddim" (MID/mid.py:333)
            x_np = x_T.numpy()[None]
            # the K samples stay on the GPU (only the k kept ones come back) while jmid_topk takes the shape; beyond its limits
            # - the reference has none - the host twin ranks them
            on_dev = k < K and self.device_topk and topk_fits_device(A, K, H)
            check = self.self_check and tuple(x_np.shape) not in self._checked_shapes
            in_cluster = None
            if (on_dev or k == K) and not check:
                # the whole call in ONE library entry (jmid_predict: encoder -> denoise -> integrator -> top-k chained on the
                # stream, one upload, one download); JMID_ERANGE -> the staged path below in exact fp32
                try:
                    out, lw = self.engine.predict(sb.x_st, sb.nbr_sum, sb.edge_mask, x_np, sb.p0[None], k, dt=self.time_step,
                                                  precision=self.precision)
                    if on_dev:
                        in_cluster, logw_in = out[0], lw[0].astype(np.float64)
                    else:
                        in_cluster = out[0].transpose(1, 0, 2, 3)                 # [K, A, H, 2] -> agents by ascending id
                        logw_in = np.log(np.ones((A, K), dtype=np.float64) / K)
                    t2 = time.perf_counter()
                except JmidError as e:
                    if e.code != -5 or self.precision == "f32":                   # JMID_ERANGE
                        raise
            if in_cluster is None:
                ctx = self.engine.encode(sb.x_st, sb.nbr_sum, sb.edge_mask)
                pos = self._denoise(x_np, ctx[None], sb.p0[None], want_pos=not on_dev or check)
                if check:        # first call of this shape in an opt-in mode: pos is the result to use (possibly f16x3's)
                    pos = self._self_check(x_np, ctx[None], sb.p0[None], pos)
                t2 = time.perf_counter()
                if on_dev:
                    # joint-KDE top-k on the device (jmid_topk) over the positions the denoise call left in the workspace
                    sel, lw = self.engine.topk(pos if check else None, k, dims=(1, A, K, H))
                    in_cluster, logw_in = sel[0], lw[0].astype(np.float64)
                elif k < K:
                    in_cluster, logw_in = most_likely_samples(pos[0], k)          # [A, k, H, 2], [A, k]: host path
                    logw_in = logw_in.astype(np.float64)
                else:
                    in_cluster = pos[0].transpose(1, 0, 2, 3)                     # [K, A, H, 2] -> agents by ascending id
                    logw_in = np.log(np.ones((A, K), dtype=np.float64) / K)
        t3 = time.perf_counter()
        forecasts = np.zeros((self.num_hums, k, H, 2), dtype=np.float64)
        logw = np.zeros((self.num_hums, k), dtype=np.float64)
        forecasts[sb.ids_in] = in_cluster
        logw[sb.ids_in] = logw_in
        for i in sb.ids_out:
            forecasts[i] = sb.cv_forecasts[int(i)][np.newaxis]
            logw[i] = logw_in[0]
        # prepend the current pose estimate (mid_sim_wrapper.py:444-454)
        pose = np.repeat(pose_now[:, None, None, :], k, axis=1)
        out = np.concatenate((pose, forecasts), axis=2), logw
        t4 = time.perf_counter()
        self.timings = {"scene_ms": 1e3 * (t1 - t0), "device_ms": 1e3 * (t2 - t1), "topk_ms": 1e3 * (t3 - t2),
                        "assemble_ms": 1e3 * (t4 - t3), "total_ms": 1e3 * (t4 - t0)}
        return out


def get_most_likely_samples(forecasts, mid_model, num_ret_samples):
    """Module-level twin of ``mid_sim_wrapper.get_most_likely_samples`` (``:14-169``; joint branch - the predictor's model has
    no ``cfg``, ``:20-21``): forecasts [K, A, H, 2] -> (kept [A, k, H, 2], log-weights [A, k]) float32 tensors.  Host arithmetic
    (``kde.most_likely_samples``); the class method of the same name uses the device kernel."""
    f = forecasts.detach().cpu().numpy() if isinstance(forecasts, torch.Tensor) else np.asarray(forecasts)
    sel, lw = most_likely_samples(np.ascontiguousarray(f, dtype=np.float32), int(num_ret_samples))
    return torch.from_numpy(np.ascontiguousarray(sel)), torch.from_numpy(np.ascontiguousarray(lw))


# size limits of the device-side joint-KDE top-k (jmid_topk, include/jmid_hip.h): beyond them the host twin (kde.py) ranks
TOPK_MAX_AGENTS, TOPK_MAX_SAMPLES, TOPK_MAX_HORIZON = 32, 1024, 24


def topk_fits_device(A: int, K: int, H: int) -> bool:
    return A <= TOPK_MAX_AGENTS and K <= TOPK_MAX_SAMPLES and H <= TOPK_MAX_HORIZON


def denoise_with_fallback(engine: JmidEngine, x_T, ctx, p0, dt: float, precision: str, want_pos: bool = True):
    """``engine.denoise`` -> positions; on JMID_ERANGE (an fp16 operand left the fp16 range) the call is repeated in the
    exact-fp32 mode - what the reference computes in throughout (diffusion.py:478-541) - and counted.  -> (pos, fell_back)."""
    global ERANGE_FALLBACKS
    try:
        _, pos = engine.denoise(x_T, ctx, p0, dt=dt, precision=precision, want_vel=False, want_pos=want_pos)
        return pos, False
    except JmidError as e:
        if e.code != -5 or precision == "f32":     # JMID_ERANGE
            raise
    ERANGE_FALLBACKS += 1
    _, pos = engine.denoise(x_T, ctx, p0, dt=dt, precision="f32", want_vel=False, want_pos=want_pos)
    return pos, True


def _engine_lock_of(engine: JmidEngine) -> RLock:
    with _CACHE_LOCK:
        return _ENGINE_LOCKS.setdefault(id(engine), RLock())


def predict_batch(engine: JmidEngine, human_xy: np.ndarray, robot_xy: np.ndarray, seeds, *, num_samples: int,
                  num_ret_samples: int, horizon: int, time_step: float, precision: Optional[str] = None,
                  device_topk: bool = True) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """``predict_ret_best()`` for E independent episodes in as few device calls as their cluster sizes allow: the feed
    of the multi-episode evaluation sweeps.

    human_xy [E, F, N, 2], robot_xy [E, F, 2] on the ``time_step`` grid (oldest first); ``seeds[e]`` seeds the torch
    generator episode e draws its x_T from (== ``torch.manual_seed(seeds[e])`` before the per-episode call).
    The host batch comes from ``scene.build_scenes_batched`` with the reference's own clustering (the pedestrians
    within 3 m of the cluster nearest the robot go through the network, mid_sim_wrapper.py:335-355; the others get
    constant-velocity forecasts, :413-429); episodes with the same number A of in-cluster pedestrians share one
    ``encode`` + ``denoise`` call (the C ABI takes one A per call).  Returns (forecasts [E, N, k, H+1, 2] float64,
    log-weights [E, N, k] float64, in_cluster [E, N] bool), each episode as ``predict_ret_best`` would return it.
    """
    E, F, N, _ = human_xy.shape
    K, k, H = int(num_samples), int(num_ret_samples), int(horizon)
    precision = DEFAULTS["precision"] if precision is None else precision      # (no self check here: an engine-level call)
    b = SC.build_scenes_batched(human_xy, robot_xy, time_step, horizon=H)
    inc = b["in_cluster"]
    forecasts = np.zeros((E, N, k, H, 2), dtype=np.float64)
    logw = np.zeros((E, N, k), dtype=np.float64)
    n_in = inc.sum(axis=1)
    for A in np.unique(n_in):
        eps = np.nonzero(n_in == A)[0]
        A = int(A)
        rows = np.stack([np.nonzero(inc[e])[0] for e in eps])                       # [Eg, A] ascending track ids
        ei = eps[:, None]
        x_T = torch.stack([torch.randn([K * A, H, 2], generator=torch.Generator().manual_seed(int(seeds[e])))
                           for e in eps]).numpy()
        p0 = np.ascontiguousarray(b["p0"][ei, rows])
        with _engine_lock_of(engine):          # an engine may be shared with forecaster instances; it is not re-entrant
            ctx = engine.encode(b["x_st"][ei, rows].reshape(len(eps) * A, F, 6),
                                b["nbr_sum"][ei, rows].reshape(len(eps) * A, 2, F, 6),
                                b["edge_mask"][ei, rows].reshape(len(eps) * A, 2)).reshape(len(eps), A, -1)
            # the samples stay on the GPU; every episode of the group in ONE jmid_topk call (host twin beyond its size limits)
            on_dev = k < K and device_topk and topk_fits_device(A, K, H)
            pos, _ = denoise_with_fallback(engine, x_T, ctx, p0, time_step, precision, want_pos=not on_dev)
            if on_dev:
                sel_all, lw_all = engine.topk(None, k, dims=(len(eps), A, K, H))   # [Eg, A, k, H, 2], [Eg, A, k]
        for g, e in enumerate(eps):                                                # pos [Eg, K, A, H, 2]
            if on_dev:
                sel, lw = sel_all[g], lw_all[g].astype(np.float64)
            elif k < K:
                sel, lw = most_likely_samples(pos[g], k)                          # [A, k, H, 2], [A, k]
                lw = lw.astype(np.float64)
            else:
                sel = pos[g].transpose(1, 0, 2, 3)
                lw = np.log(np.ones((A, K), dtype=np.float64) / K)
            forecasts[e, rows[g]] = sel
            logw[e, rows[g]] = lw
            out_rows = np.nonzero(~inc[e])[0]
            forecasts[e, out_rows] = b["cv"][e, out_rows][:, None]
            logw[e, out_rows] = lw[0]
    pose = np.repeat(human_xy[:, -1][:, :, None, None, :], k, axis=2)              # [E, N, k, 1, 2]
    return np.concatenate((pose, forecasts), axis=3), logw, inc


def write_configs(directory: str, *, joint: bool, ctx_dim: int, N: int, K: int, k_ret: int, H: int, step: int,
                  past: int = 6, time_step: float = 0.25, model_path: str = "weights.npz"):
    """Helper for tests / demos: writes an env.config + MID yaml pair with the keys the predictor reads
    (``configs/env.config:8-13``, ``JMID/test_time_configs/mid_jp.yaml``) and returns (RawConfigParser, yaml path)."""
    os.makedirs(directory, exist_ok=True)
    cfg = dict(model_path=model_path,
               diffnet="JointPredictionTransformerConcatLinear" if joint else "TransformerConcatLinear",
               encoder_dim=ctx_dim, tf_layer=3, num_samples=K, step_size=step, prediction_horizon=H,
               maximum_history_length=past - 1, sampling="ddim", eval_mode=True, time=False,
               override_attention_radius=[])
    ypath = os.path.join(directory, "mid_jp.yaml" if joint else "mid.yaml")
    with open(ypath, "w") as f:
        yaml.safe_dump(cfg, f)
    env = configparser.RawConfigParser()
    env.add_section("env")
    env.set("env", "time_step", str(time_step))
    env.add_section("human_trajectory_forecaster")
    env.set("human_trajectory_forecaster", "publish_freq", "0.08")
    env.set("human_trajectory_forecaster", "past_num_frames", str(past))
    env.set("human_trajectory_forecaster", "prediction_horizon", str(H))
    env.set("human_trajectory_forecaster", "num_samples", str(k_ret))
    env.add_section("sim")
    env.set("sim", "human_num", str(N))
    return env, ypath
