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

Arithmetic (``precision``): the contractions run in the library's F16MX mode by default - fp16 activation operand times
split-fp16 weight, as ``A_hi . W_hi`` on the fp16 matrix cores plus the correction term ``A_hi . W_lo`` as ONE bf8 x bf8
MFMA per 64-deep block (1.5 MFMA passes per GEMM product), fp32 accumulation; three-term softmax logits (their correction
terms on the same fp8 path), one fp16 plane of the attention weights, residual stream, LayerNorm and DDIM state at
fp32-class precision.  It is the mode ``bench.py`` quotes, >= the bf16 BASELINE.json names for this
workload, and it holds every reference golden fixture inside the 1e-4 m mean-ADE gate with the same errors as
``precision="f16x2"`` (the same products with the correction terms in fp16: two passes; worst 5.7e-5 m on the 2-step
fixtures, 5e-6 m on the 50-step cfg3 sample).  ``"f16x3"`` selects the fp32-class three-term
products (mean ADE ~1e-6 m, the fp32-vs-fp64 noise floor; ~40 % fewer trajectories per second on batches), ``"f32"`` the
exact-fp32 MFMA path.  If an activation ever leaves the fp16 range the call is repeated transparently in the exact-fp32 mode.

RNG contract (``rng_compat``): ``x_T`` is always the first draw of torch's CPU default generator
(``MID/models/diffusion.py:499``).  The reference also draws one (unused, DDIM) ``randn_like(x_T)`` per reverse step
(``:509``) on the device ``x_T`` was moved to: the CPU generator in a CPU-only run (what the golden captures are),
the CUDA generator on a GPU host (``MID/mid.py:91``).  ``rng_compat="cpu"`` consumes those draws from the CPU
generator, ``"cuda"`` from the device generator (the CPU generator then advances by ``x_T`` only), and the default
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
        if key in _WEIGHTS_CACHE:
            return _WEIGHTS_CACHE[key]
        if p.endswith(".npz"):
            w = JMIDWeights.load(p)
            if w.dims != dims:
                raise ValueError(f"{p} holds dims {w.dims}, config asks for {dims}")
        else:
            ckpt = torch.load(p, map_location="cpu", weights_only=False)
            w = JMIDWeights.from_reference_state(dims, ckpt["ddpm"], ckpt["encoder"])
        _WEIGHTS_CACHE[key] = w
        return w
    raise FileNotFoundError(f"no checkpoint at {model_path} (or its .npz export)")


class ForecasterSimSuper:
    """History bookkeeping, identical to ``mid_sim_wrapper.py:172-204``."""

    def init_super(self, env_config):
        self.prev_states_lock = Lock()
        if env_config is None:     # the reference falls back to a config file that is not in its tree and then fails here
            raise configparser.NoSectionError("human_trajectory_forecaster")
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
                 device_id: int = 0, precision: str = "f16mx", rng_compat: str = "auto"):
        self.init_super(env_config)
        self.precision = precision
        if rng_compat not in ("auto", "cpu", "cuda"):
            raise ValueError("rng_compat must be 'auto', 'cpu' or 'cuda'")
        self.rng_compat = rng_compat if rng_compat != "auto" else ("cuda" if torch.cuda.is_available() else "cpu")
        self._init_MID(mid_config_file, weights, device_id)

    def _init_MID(self, mid_config_file, weights, device_id):
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
        key = (weights.checksum(), self.joint, device_id, self.num_hist_frames, self.step_size)
        eng = _ENGINE_CACHE.get(key)
        if eng is None:
            eng = JmidEngine(weights, joint=self.joint, device_id=device_id, hist_len=self.num_hist_frames,
                             step=self.step_size)
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

    def predict_ret_best(self) -> Tuple[np.ndarray, np.ndarray]:
        """mid_sim_wrapper.py:482-510 -> (forecasts [N, k, H+1, 2] float64, log-weights [N, k] float64)."""
        prev, rob = self._snapshot()
        hum_xy, rob_xy, pose_now = SC.frame_table(prev, rob, self.time_step, self.num_hist_frames)
        sb = SC.build_scene(hum_xy, rob_xy, self.time_step, self.predict_horizon, self.num_hist_frames)
        A, K, H, k = len(sb.ids_in), self.num_samples, self.predict_horizon, self.num_ret_samples
        # RNG contract (module docstring): x_T is the first draw of the CPU default generator; the per-step z of the
        # reference (unused by DDIM) comes from the generator of the device the reference would run on
        x_T = torch.randn([K * A, H, 2])
        stride = int(100 / self.step_size)
        z_like = x_T if self.rng_compat == "cpu" else torch.empty_like(x_T, device=f"cuda:{self.engine.device_id}")
        for t in range(100, 0, -stride):
            if t > 1:
                torch.randn_like(z_like)
        with self._engine_lock:               # the engine is shared between forecaster instances and not re-entrant
            if self.engine.step != self.step_size or self.engine.sampling != "ddim":
                self.engine.set_step(self.step_size, "ddim")   # eval_sicnav hard-codes sampling="ddim" (MID/mid.py:333)
            ctx = self.engine.encode(sb.x_st, sb.nbr_sum, sb.edge_mask)
            try:
                _, pos = self.engine.denoise(x_T.numpy()[None], ctx[None], sb.p0[None], dt=self.time_step,
                                             precision=self.precision, want_vel=False)
            except JmidError as e:
                if e.code != -5 or self.precision == "f32":     # JMID_ERANGE: an operand left the fp16 range
                    raise
                _, pos = self.engine.denoise(x_T.numpy()[None], ctx[None], sb.p0[None], dt=self.time_step,
                                             precision="f32", want_vel=False)   # exact-fp32 MFMA path, same result
        samples = pos[0]                                                  # [K, A, H, 2], agents by ascending id
        if k < K:
            in_cluster, logw_in = most_likely_samples(samples, k)        # [A, k, H, 2], [A, k]
            logw_in = logw_in.astype(np.float64)
        else:
            in_cluster = samples.transpose(1, 0, 2, 3)
            logw_in = np.log(np.ones((A, K), dtype=np.float64) / K)
        forecasts = np.zeros((self.num_hums, k, H, 2), dtype=np.float64)
        logw = np.zeros((self.num_hums, k), dtype=np.float64)
        forecasts[sb.ids_in] = in_cluster
        logw[sb.ids_in] = logw_in
        for i in sb.ids_out:
            forecasts[i] = sb.cv_forecasts[int(i)][np.newaxis]
            logw[i] = logw_in[0]
        # prepend the current pose estimate (mid_sim_wrapper.py:444-454)
        pose = np.repeat(pose_now[:, None, None, :], k, axis=1)
        return np.concatenate((pose, forecasts), axis=2), logw


def predict_batch(engine: JmidEngine, human_xy: np.ndarray, robot_xy: np.ndarray, seeds, *, num_samples: int,
                  num_ret_samples: int, horizon: int, time_step: float, precision: str = "f16mx"
                  ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
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
        ctx = engine.encode(b["x_st"][ei, rows].reshape(len(eps) * A, F, 6),
                            b["nbr_sum"][ei, rows].reshape(len(eps) * A, 2, F, 6),
                            b["edge_mask"][ei, rows].reshape(len(eps) * A, 2)).reshape(len(eps), A, -1)
        p0 = np.ascontiguousarray(b["p0"][ei, rows])
        try:
            _, pos = engine.denoise(x_T, ctx, p0, dt=time_step, precision=precision, want_vel=False)
        except JmidError as e:
            if e.code != -5 or precision == "f32":
                raise
            _, pos = engine.denoise(x_T, ctx, p0, dt=time_step, precision="f32", want_vel=False)
        for g, e in enumerate(eps):                                                # pos [Eg, K, A, H, 2]
            if k < K:
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
