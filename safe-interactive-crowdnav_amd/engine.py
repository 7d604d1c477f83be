"""JmidEngine: Python handle around the C ABI (``include/jmid_hip.h``).

This is the device half of the predictor: context encoder, the batched DDIM
reverse-denoising loop and the integrator, i.e. what ``AutoEncoder.generate_sicnav_inference``
(``sicnav_diffusion/JMID/MID/models/autoencoder.py:17-47``) does, for one scene or for a
batch of independent episodes.  No arithmetic happens in Python.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple, Union

import numpy as np

from . import _lib
from ._lib import JmidError
from .schedule import VarianceSchedule, ddim_steps, ddpm_steps
from .weights import JMIDWeights

try:  # torch is optional plumbing here (device tensors in / out)
    import torch
except Exception:  # pragma: no cover
    torch = None

ArrayLike = Union[np.ndarray, "torch.Tensor"]

# every JMID_ERANGE any engine of this process has returned (the tests assert that no fixture ever produces one)
ERANGE_EVENTS: list = []
# every JMID_ETIMEOUT (a workgroup of a one-launch GEMM + LayerNorm gave up waiting for its partners: include/jmid_hip.h) - the engine
# repeats such a call once in the SAME precision (the handle has dropped that kernel by then) and records it here
TIMEOUT_EVENTS: list = []


def _is_cuda(t) -> bool:
    return torch is not None and isinstance(t, torch.Tensor) and t.is_cuda


class _Buf:
    """fp32 contiguous view of an input + its raw pointer."""

    def __init__(self, a: ArrayLike, device_mode: bool):
        if device_mode:
            if not _is_cuda(a):
                raise TypeError("device-mode call needs CUDA/HIP torch tensors for every array argument")
            self.keep = a.detach().to(torch.float32).contiguous()
            self.ptr = C.c_void_p(self.keep.data_ptr())
        else:
            if torch is not None and isinstance(a, torch.Tensor):
                a = a.detach().cpu().numpy()
            self.keep = np.ascontiguousarray(a, dtype=np.float32)
            self.ptr = C.c_void_p(self.keep.ctypes.data)


class JmidEngine:
    """One predictor engine (weights resident on one GPU, one HIP stream)."""

    def __init__(self, weights: JMIDWeights, joint: bool, device_id: int = 0, hist_len: int = 6,
                 step: int = 50, schedule: Optional[VarianceSchedule] = None, lib_path: Optional[str] = None):
        self._lib = _lib.load_library(lib_path)      # lib_path: another build of the library (tests: both flavours in one process)
        self._h = _lib.Handle()
        self.dims = weights.dims
        self.joint = bool(joint)
        self.device_id = device_id
        self._caller_stream = 0          # NULL: the legacy default stream
        self.hist_len = hist_len
        rc = self._lib.jmid_create(C.byref(self._h), device_id, _lib.NET_JMID if joint else _lib.NET_IMID,
                                   self.dims.ctx_dim, self.dims.tf_layer, self.dims.nhead, hist_len)
        if rc != 0:
            msg = self._lib.jmid_last_error(None).decode()
            self._h = None
            raise JmidError(rc, msg)
        for name, t in weights.tensors.items():
            a = np.ascontiguousarray(t.numpy(), dtype=np.float32)
            self._check(self._lib.jmid_load_weight(self._h, name.encode(), C.c_void_p(a.ctypes.data), a.size))
        self._check(self._lib.jmid_finalize_weights(self._h))
        self.schedule = schedule or VarianceSchedule.linear()
        self.set_step(step)

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc: int) -> None:
        if rc == -5:
            ERANGE_EVENTS.append((id(self), self._lib.jmid_last_error(self._h).decode()))
        if rc != 0:
            raise JmidError(rc, self._lib.jmid_last_error(self._h).decode())

    def _compute(self, fn, *args) -> None:
        """A compute entry of the C ABI.  JMID_ETIMEOUT is not an arithmetic condition: the library has switched this handle to the
        unfused kernels (same bits), so the call is repeated once, as it is, and the event recorded."""
        rc = fn(*args)
        if rc == -6:
            TIMEOUT_EVENTS.append((id(self), self._lib.jmid_last_error(self._h).decode()))
            rc = fn(*args)
        self._check(rc)

    def timeout_count(self) -> int:
        """Calls on this engine that ended with JMID_ETIMEOUT (at most one in practice: the first switches the handle for good)."""
        return int(self._lib.jmid_timeout_count(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.jmid_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def set_step(self, step: int, sampling: str = "ddim", flexibility: float = 0.0) -> None:
        """``step`` = the reference's ``step_size`` yaml key: number of reverse iterations out of 100
        (stride = int(100/step), MID/models/diffusion.py:507); ``sampling`` = "ddim" (what the predictor uses,
        MID/mid.py:333) or "ddpm" with ``flexibility`` (``get_sigmas``, diffusion.py:59-64)."""
        self.step, self.sampling = step, sampling
        if sampling == "ddim":
            tab = ddim_steps(self.schedule, step)
            cols = [np.array([getattr(s, k) for s in tab], dtype=np.float32) for k in ("beta", "c_e", "c_x", "n_x", "n_e")]
            self._check(self._lib.jmid_set_ddim_table(self._h, len(tab), *[C.c_void_p(c.ctypes.data) for c in cols]))
        elif sampling == "ddpm":
            tab = ddpm_steps(self.schedule, step, flexibility)
            cols = [np.array([getattr(s, k) for s in tab], dtype=np.float32) for k in ("beta", "c0", "c1", "sigma")]
            cols.append(np.array([int(s.noise) for s in tab], dtype=np.int32))
            self._check(self._lib.jmid_set_ddpm_table(self._h, len(tab), *[C.c_void_p(c.ctypes.data) for c in cols]))
        else:
            raise ValueError("sampling must be 'ddim' or 'ddpm'")
        self.n_steps = len(tab)

    def set_chunk_episodes(self, n: int) -> None:
        self._check(self._lib.jmid_set_chunk_episodes(self._h, int(n)))

    def set_tuning(self, key: str, value: int) -> None:
        self._check(self._lib.jmid_set_tuning(self._h, key.encode(), int(value)))

    def _mem(self, dev: bool) -> int:
        """Memory mode of a call; device-mode calls are ordered against torch's CURRENT stream (include/jmid_hip.h)."""
        if not dev:
            return _lib.MEM_HOST
        st = int(torch.cuda.current_stream(self.device_id).cuda_stream)
        if st != self._caller_stream:
            self._check(self._lib.jmid_set_caller_stream(self._h, C.c_void_p(st)))
            self._caller_stream = st
        return _lib.MEM_DEVICE

    def graph_replays(self) -> int:
        """Calls whose denoise loop ran as a replayed hipGraph (small one-chunk calls from their third repetition on)."""
        return int(self._lib.jmid_graph_replays(self._h))

    def erange_count(self) -> int:
        """Calls on this engine that ended with JMID_ERANGE (an activation left the fp16 range: the caller repeats in "f32")."""
        return int(self._lib.jmid_erange_count(self._h))

    def synchronize(self) -> None:
        self._check(self._lib.jmid_synchronize(self._h))

    # ------------------------------------------------------------------ compute
    def encode(self, x_st: ArrayLike, nbr_sum: ArrayLike, edge_mask: ArrayLike) -> ArrayLike:
        """x_st [n, hist, 6], nbr_sum [n, 2, hist, 6], edge_mask [n, 2] -> ctx [n, ctx_dim]."""
        dev = _is_cuda(x_st)
        n = int(x_st.shape[0])
        if tuple(x_st.shape) != (n, self.hist_len, 6) or tuple(nbr_sum.shape) != (n, 2, self.hist_len, 6) \
                or tuple(edge_mask.shape) != (n, 2):
            raise ValueError("bad encoder input shapes")
        a, b, c = _Buf(x_st, dev), _Buf(nbr_sum, dev), _Buf(edge_mask, dev)
        if dev:
            out = torch.empty((n, self.dims.ctx_dim), dtype=torch.float32, device=x_st.device)
            optr = C.c_void_p(out.data_ptr())
        else:
            out = np.empty((n, self.dims.ctx_dim), dtype=np.float32)
            optr = C.c_void_p(out.ctypes.data)
        self._check(self._lib.jmid_encode(self._h, n, a.ptr, b.ptr, c.ptr, optr,
                                          self._mem(dev)))
        return out

    def _shapes(self, x, ctx) -> Tuple[int, int, int, int]:
        if x.ndim != 4 or ctx.ndim != 3 or x.shape[0] != ctx.shape[0] or x.shape[-1] != 2:
            raise ValueError("expected x [E, K*A, T, 2] and ctx [E, A, ctx_dim]")
        E, KA, T, _ = (int(v) for v in x.shape)
        A = int(ctx.shape[1])
        if int(ctx.shape[2]) != self.dims.ctx_dim or KA % A != 0:
            raise ValueError("ctx width / row count mismatch")
        return E, A, KA // A, T

    def denoise(self, x_T: ArrayLike, ctx: ArrayLike, p0: Optional[ArrayLike] = None, dt: float = 0.25,
                precision: str = "f32", want_vel: bool = True, want_pos: bool = True, z: Optional[ArrayLike] = None):
        """Batched reverse-denoising loop.  x_T [E, K*A, T, 2], ctx [E, A, ctx_dim], p0 [E, A, 2].
        ``z`` [n_steps, E, K*A, T, 2]: per-step normal draws, required when the DDPM table is installed.
        Returns (vel [E,K,A,T,2] or None, pos [E,K,A,T,2] or None)."""
        dev = _is_cuda(x_T)
        E, A, K, T = self._shapes(x_T, ctx)
        want_pos = want_pos and p0 is not None
        bx, bc = _Buf(x_T, dev), _Buf(ctx, dev)
        bp = _Buf(p0, dev) if p0 is not None else None
        shape = (E, K, A, T, 2)

        def alloc():
            if dev:
                t = torch.empty(shape, dtype=torch.float32, device=x_T.device)
                return t, C.c_void_p(t.data_ptr())
            t = np.empty(shape, dtype=np.float32)
            return t, C.c_void_p(t.ctypes.data)

        vel, vptr = alloc() if want_vel else (None, None)
        pos, pptr = alloc() if want_pos else (None, None)
        if z is not None:
            if tuple(z.shape) != (self.n_steps, E, K * A, T, 2):
                raise ValueError("z must be [n_steps, E, K*A, T, 2]")
            bz = _Buf(z, dev)
            self._compute(self._lib.jmid_denoise_ddpm, self._h, E, A, K, T, bx.ptr, bz.ptr, bc.ptr, bp.ptr if bp else None,
                                                    float(dt), _lib.PRECISIONS[precision], vptr, pptr,
                                                    self._mem(dev))
        else:
            self._compute(self._lib.jmid_denoise, self._h, E, A, K, T, bx.ptr, bc.ptr, bp.ptr if bp else None, float(dt),
                                               _lib.PRECISIONS[precision], vptr, pptr,
                                               self._mem(dev))
        return vel, pos

    def topk(self, pos: Optional[ArrayLike], k: int, dims: Optional[Tuple[int, int, int, int]] = None):
        """Joint-KDE top-k on the device (``jmid_topk``; get_most_likely_samples, mid_sim_wrapper.py:14-169), batched over
        episodes.  pos [E, K, A, T, 2] -> (kept [E, A, k, T, 2], log-weights [E, A, k]) in ascending likelihood.
        ``pos=None`` with ``dims=(E, A, K, T)`` ranks the positions of the preceding ``denoise`` call, which are still in the
        engine's workspace (nothing but the k kept samples comes back to the host)."""
        import math
        if pos is None:
            if dims is None:
                raise ValueError("pos=None needs dims=(E, A, K, T) of the preceding denoise call")
            E, A, K, T = (int(v) for v in dims)
            dev, bp = False, None
        else:
            dev = _is_cuda(pos)
            E, K, A, T, _ = (int(v) for v in pos.shape)
            bp = _Buf(pos, dev)
        # the bandwidths exactly as the reference computes them (fp32 torch ops, mid_sim_wrapper.py:26-30)
        bw = torch.exp(torch.linspace(math.log(0.01), math.log(0.1), steps=T))
        if dev:
            bwb = bw.to(pos.device)
            sel = torch.empty((E, A, k, T, 2), dtype=torch.float32, device=pos.device)
            lw = torch.empty((E, A, k), dtype=torch.float32, device=pos.device)
            ptrs = (C.c_void_p(bwb.data_ptr()), C.c_void_p(sel.data_ptr()), C.c_void_p(lw.data_ptr()))
        else:
            bwb = np.ascontiguousarray(bw.numpy())
            sel = np.empty((E, A, k, T, 2), dtype=np.float32)
            lw = np.empty((E, A, k), dtype=np.float32)
            ptrs = (C.c_void_p(bwb.ctypes.data), C.c_void_p(sel.ctypes.data), C.c_void_p(lw.ctypes.data))
        self._check(self._lib.jmid_topk(self._h, E, A, K, T, int(k), bp.ptr if bp is not None else None, *ptrs,
                                        self._mem(dev)))
        return sel, lw

    def predict(self, x_st: np.ndarray, nbr_sum: np.ndarray, edge_mask: np.ndarray, x_T: np.ndarray, p0: np.ndarray, k: int,
                dt: float = 0.25, precision: str = "f32"):
        """One predictor call end to end (``jmid_predict``): encoder -> denoise loop -> integrator -> joint-KDE top-k, host arrays in
        and out, one upload, one download, nothing in between.  x_st [E*A, hist, 6], nbr_sum [E*A, 2, hist, 6], edge_mask [E*A, 2],
        x_T [E, K*A, T, 2], p0 [E, A, 2].  k < K -> (kept [E, A, k, T, 2], log-weights [E, A, k]); k == K -> (pos [E, K, A, T, 2], None)."""
        import math
        E, KA, T, _ = (int(v) for v in x_T.shape)
        A = int(p0.shape[1])
        K = KA // A
        if tuple(x_st.shape) != (E * A, self.hist_len, 6) or tuple(nbr_sum.shape) != (E * A, 2, self.hist_len, 6) \
                or tuple(edge_mask.shape) != (E * A, 2) or KA != K * A or tuple(p0.shape) != (E, A, 2):
            raise ValueError("bad predict() input shapes")
        b = [_Buf(a, False) for a in (x_st, nbr_sum, edge_mask, x_T, p0)]
        if k < K:
            bw = np.ascontiguousarray(torch.exp(torch.linspace(math.log(0.01), math.log(0.1), steps=T)).numpy())   # mid_sim_wrapper.py:26-30
            sel = np.empty((E, A, k, T, 2), dtype=np.float32)
            lw = np.empty((E, A, k), dtype=np.float32)
            self._compute(self._lib.jmid_predict, self._h, E, A, K, T, int(k), *[x.ptr for x in b], float(dt), _lib.PRECISIONS[precision],
                                               C.c_void_p(bw.ctypes.data), C.c_void_p(sel.ctypes.data), C.c_void_p(lw.ctypes.data), None)
            return sel, lw
        pos = np.empty((E, K, A, T, 2), dtype=np.float32)
        self._compute(self._lib.jmid_predict, self._h, E, A, K, T, int(k), *[x.ptr for x in b], float(dt), _lib.PRECISIONS[precision],
                                           None, None, None, C.c_void_p(pos.ctypes.data))
        return pos, None

    def net_eval(self, x: ArrayLike, ctx: ArrayLike, step_idx: int = 0, precision: str = "f32"):
        """One evaluation of e_theta for DDIM table entry ``step_idx``; x [E, K*A, T, 2] -> e same shape."""
        dev = _is_cuda(x)
        E, A, K, T = self._shapes(x, ctx)
        bx, bc = _Buf(x, dev), _Buf(ctx, dev)
        if dev:
            out = torch.empty(tuple(x.shape), dtype=torch.float32, device=x.device)
            optr = C.c_void_p(out.data_ptr())
        else:
            out = np.empty(tuple(x.shape), dtype=np.float32)
            optr = C.c_void_p(out.ctypes.data)
        self._compute(self._lib.jmid_net_eval, self._h, E, A, K, T, int(step_idx), bx.ptr, bc.ptr,
                                            _lib.PRECISIONS[precision], optr,
                                            self._mem(dev))
        return out

    def episode_metrics(self, pos: ArrayLike, gt: ArrayLike) -> ArrayLike:
        """pos [E,K,A,T,2], gt [E,A,T,2] -> [E,4] = (mean ADE, joint min ADE, mean FDE, joint min FDE)."""
        dev = _is_cuda(pos)
        E, K, A, T, _ = (int(v) for v in pos.shape)
        if tuple(gt.shape) != (E, A, T, 2):
            raise ValueError("gt must be [E, A, T, 2]")
        bp, bg = _Buf(pos, dev), _Buf(gt, dev)
        if dev:
            out = torch.empty((E, 4), dtype=torch.float32, device=pos.device)
            optr = C.c_void_p(out.data_ptr())
        else:
            out = np.empty((E, 4), dtype=np.float32)
            optr = C.c_void_p(out.ctypes.data)
        self._check(self._lib.jmid_episode_metrics(self._h, E, A, K, T, bp.ptr, bg.ptr, optr,
                                                   self._mem(dev)))
        return out

    # ------------------------------------------------------------------ measurement
    def kernel_classes(self):
        return [self._lib.jmid_kernel_class_name(i).decode() for i in range(self._lib.jmid_kernel_class_count())]

    def profile_enable(self, classes=None) -> None:
        names = self.kernel_classes()
        mask = 0
        for i, n in enumerate(names):
            if classes is None or n in classes:
                mask |= 1 << i
        self._check(self._lib.jmid_profile_enable(self._h, mask))

    def profile_disable(self) -> None:
        self._check(self._lib.jmid_profile_enable(self._h, 0))

    def profile_reset(self) -> None:
        self._check(self._lib.jmid_profile_reset(self._h))

    def profile_get(self) -> Dict[str, Tuple[int, float]]:
        out = {}
        for i, n in enumerate(self.kernel_classes()):
            cnt, ms = C.c_int64(0), C.c_double(0.0)
            self._check(self._lib.jmid_profile_get(self._h, i, C.byref(cnt), C.byref(ms)))
            out[n] = (int(cnt.value), float(ms.value))
        return out

    # ------------------------------------------------------------------ diagnostics (unit tests)
    def dbg_gemm(self, A: np.ndarray, Wt: np.ndarray, bias: Optional[np.ndarray], relu: bool = False,
                 precision: str = "f32") -> np.ndarray:
        A = np.ascontiguousarray(A, np.float32)
        Wt = np.ascontiguousarray(Wt, np.float32)
        M, K = A.shape
        N = Wt.shape[0]
        out = np.empty((M, N), np.float32)
        b = np.ascontiguousarray(bias, np.float32) if bias is not None else None
        self._check(self._lib.jmid_dbg_gemm(self._h, M, N, K, C.c_void_p(A.ctypes.data), C.c_void_p(Wt.ctypes.data),
                                            C.c_void_p(b.ctypes.data) if b is not None else None, int(relu),
                                            _lib.PRECISIONS[precision], C.c_void_p(out.ctypes.data)))
        return out

    def dbg_attention(self, qkv: np.ndarray, nseq: int, S: int, precision: str = "f32") -> np.ndarray:
        qkv = np.ascontiguousarray(qkv, np.float32)
        d = 2 * self.dims.ctx_dim
        assert qkv.shape == (nseq * S, 3 * d)
        out = np.empty((nseq * S, d), np.float32)
        self._check(self._lib.jmid_dbg_attention(self._h, nseq, S, C.c_void_p(qkv.ctypes.data),
                                                 _lib.PRECISIONS[precision], C.c_void_p(out.ctypes.data)))
        return out

    def dbg_gemm_ln_mx(self, A: np.ndarray, Wt: np.ndarray, bias, gamma, beta, X: np.ndarray, fused) -> np.ndarray:
        """LayerNorm(X + A Wt^T + bias) with the F16MX second-generation kernels (d_model 512): fused = 1 the row-complete kernel,
        0 the GEMM + add_ln2 pair, 3 the small-launch GEMM whose workgroups exchange the row statistics and normalise their own
        columns (gemm_small.hpp, OUT_LNX)."""
        A = np.ascontiguousarray(A, np.float32)
        Wt = np.ascontiguousarray(Wt, np.float32)
        X = np.array(X, np.float32, order="C", copy=True)
        v = [np.ascontiguousarray(t, np.float32) for t in (bias, gamma, beta)]
        M, K = A.shape
        assert Wt.shape == (512, K) and X.shape == (M, 512)
        self._check(self._lib.jmid_dbg_gemm_ln_mx(self._h, M, K, C.c_void_p(A.ctypes.data), C.c_void_p(Wt.ctypes.data),
                                                  *[C.c_void_p(t.ctypes.data) for t in v], C.c_void_p(X.ctypes.data),
                                                  int(fused)))
        return X

    def dbg_add_layernorm(self, X: np.ndarray, Y: np.ndarray, gamma: np.ndarray, beta: np.ndarray) -> np.ndarray:
        X = np.array(X, np.float32, order="C", copy=True)
        Y = np.ascontiguousarray(Y, np.float32)
        g = np.ascontiguousarray(gamma, np.float32)
        b = np.ascontiguousarray(beta, np.float32)
        M, d = X.shape
        self._check(self._lib.jmid_dbg_add_layernorm(self._h, M, d, C.c_void_p(X.ctypes.data),
                                                     C.c_void_p(Y.ctypes.data), C.c_void_p(g.ctypes.data),
                                                     C.c_void_p(b.ctypes.data)))
        return X
