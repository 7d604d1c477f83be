"""Weight container for the JMID / iMID predictor.

PyTorch is used here only as a *container* (CPU fp32 tensors keyed by the
reference's own parameter names); all arithmetic happens in the HIP library.

Names follow the reference checkpoint (``sicnav_diffusion/JMID/MID/mid.py:1231-1232,
1291, 1501-1509``): the diffusion net's keys are those of
``DiffusionTraj.net.state_dict()`` (``MID/models/diffusion.py:112-171``) and the
context-encoder keys are ``"<registrar module name>.<param>"`` for the five
modules that are live at inference (``MID/models/encoders/mgcvae.py:99-106,
189-204, 381-388``).  The template copy ``layer.*`` that the reference saves but
never executes (``diffusion.py:161``) and the ``pos_emb.pe`` buffer (recomputed)
are not part of the container.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Mapping

import numpy as np
import torch

NODE_HIST = "PEDESTRIAN/node_history_encoder"
EDGE_PED = "PEDESTRIAN->PEDESTRIAN/edge_encoder"
EDGE_ROBOT = "PEDESTRIAN->JRDB_ROBOT/edge_encoder"
EDGE_INFL = "PEDESTRIAN/edge_influence_encoder"
STATE_LEN = 6  # [px, py, vx, vy, ax, ay] (MID/utils/trajectron_hypers.py:56-61)


@dataclass(frozen=True)
class NetDims:
    """Dimensions of the denoising net, derived the way the reference ctor does
    (``diffusion.py:113-131``: d_model = 2*context_dim, nhead = 4, ff = 4*context_dim)."""

    ctx_dim: int = 256
    tf_layer: int = 3
    nhead: int = 4

    @property
    def d_model(self) -> int:
        return 2 * self.ctx_dim

    @property
    def d_ff(self) -> int:
        return 4 * self.ctx_dim

    @property
    def d_mid(self) -> int:  # concat3 output
        return self.ctx_dim

    @property
    def d_low(self) -> int:  # concat4 output
        return self.ctx_dim // 2

    @property
    def d_cond(self) -> int:  # [beta, sin, cos] + ctx
        return self.ctx_dim + 3

    @property
    def enc_hidden(self) -> int:  # LSTM hidden = encoder_dim // 2 (mid.py:1216-1219)
        return self.ctx_dim // 2


def _csl_shapes(prefix: str, d_in: int, d_out: int, d_cond: int) -> "OrderedDict[str, tuple]":
    # ConcatSquashLinear (MID/models/common.py:58-72)
    return OrderedDict(
        [
            (f"{prefix}._layer.weight", (d_out, d_in)),
            (f"{prefix}._layer.bias", (d_out,)),
            (f"{prefix}._hyper_bias.weight", (d_out, d_cond)),
            (f"{prefix}._hyper_gate.weight", (d_out, d_cond)),
            (f"{prefix}._hyper_gate.bias", (d_out,)),
        ]
    )


def diffnet_shapes(dims: NetDims) -> "OrderedDict[str, tuple]":
    d, ff, c = dims.d_model, dims.d_ff, dims.d_cond
    out = OrderedDict()
    out.update(_csl_shapes("concat1", 2, d, c))
    for l in range(dims.tf_layer):
        p = f"transformer_encoder.layers.{l}"
        out[f"{p}.self_attn.in_proj_weight"] = (3 * d, d)
        out[f"{p}.self_attn.in_proj_bias"] = (3 * d,)
        out[f"{p}.self_attn.out_proj.weight"] = (d, d)
        out[f"{p}.self_attn.out_proj.bias"] = (d,)
        out[f"{p}.linear1.weight"] = (ff, d)
        out[f"{p}.linear1.bias"] = (ff,)
        out[f"{p}.linear2.weight"] = (d, ff)
        out[f"{p}.linear2.bias"] = (d,)
        out[f"{p}.norm1.weight"] = (d,)
        out[f"{p}.norm1.bias"] = (d,)
        out[f"{p}.norm2.weight"] = (d,)
        out[f"{p}.norm2.bias"] = (d,)
    out.update(_csl_shapes("concat3", d, dims.d_mid, c))
    out.update(_csl_shapes("concat4", dims.d_mid, dims.d_low, c))
    out.update(_csl_shapes("linear", dims.d_low, 2, c))
    return out


def encoder_shapes(dims: NetDims) -> "OrderedDict[str, tuple]":
    h = dims.enc_hidden
    out = OrderedDict()
    for name, d_in in ((NODE_HIST, STATE_LEN), (EDGE_PED, 2 * STATE_LEN), (EDGE_ROBOT, 2 * STATE_LEN)):
        out[f"{name}.weight_ih_l0"] = (4 * h, d_in)
        out[f"{name}.weight_hh_l0"] = (4 * h, h)
        out[f"{name}.bias_ih_l0"] = (4 * h,)
        out[f"{name}.bias_hh_l0"] = (4 * h,)
    # AdditiveAttention(enc=h, dec=h) -> internal dim h (components/additive_attention.py:10-22)
    out[f"{EDGE_INFL}.w1.weight"] = (h, h)
    out[f"{EDGE_INFL}.w2.weight"] = (h, h)
    out[f"{EDGE_INFL}.v.weight"] = (1, h)
    return out


def all_shapes(dims: NetDims) -> "OrderedDict[str, tuple]":
    out = diffnet_shapes(dims)
    out.update(encoder_shapes(dims))
    return out


class JMIDWeights:
    """Flat name -> CPU fp32 tensor container (diffusion net + live encoder modules)."""

    def __init__(self, dims: NetDims, tensors: Mapping[str, torch.Tensor]):
        self.dims = dims
        shapes = all_shapes(dims)
        missing = [k for k in shapes if k not in tensors]
        if missing:
            raise KeyError(f"missing weights: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        self.tensors: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for k, shp in shapes.items():
            t = torch.as_tensor(tensors[k]).detach().to(torch.float32).contiguous().cpu()
            if tuple(t.shape) != tuple(shp):
                raise ValueError(f"{k}: expected shape {shp}, got {tuple(t.shape)}")
            self.tensors[k] = t

    # ------------------------------------------------------------------ builders
    @classmethod
    def from_seed(cls, dims: NetDims, seed: int) -> "JMIDWeights":
        """Deterministic synthetic weights.

        Uses numpy's PCG64 (bit-stable across numpy versions and machines), *not*
        torch's default initialisers, so that the build container and the GPU box
        regenerate identical tensors from (dims, seed): U(-1/sqrt(fan_in), 1/sqrt(fan_in))
        for matrices and biases, LayerNorm affine = 1 + 0.1 U(-1,1) / 0.1 U(-1,1).
        """
        rng = np.random.default_rng(seed)
        out = OrderedDict()
        for k, shp in all_shapes(dims).items():
            if ".norm" in k and k.endswith(".weight"):
                a = 1.0 + 0.1 * rng.uniform(-1.0, 1.0, shp)
            elif ".norm" in k and k.endswith(".bias"):
                a = 0.1 * rng.uniform(-1.0, 1.0, shp)
            else:
                fan_in = shp[-1] if len(shp) == 2 else shp[0]
                if len(shp) == 1 and "bias" in k:
                    # bias bound follows the matching weight's fan-in the way nn.Linear/nn.LSTM do;
                    # use the output width as a stand-in (only the scale matters for synthetic data)
                    fan_in = max(shp[0] // 4, 1) if "_l0" in k else shp[0]
                b = 1.0 / math.sqrt(fan_in)
                a = rng.uniform(-b, b, shp)
            out[k] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        return cls(dims, out)

    @classmethod
    def from_reference_state(cls, dims: NetDims, net_state: Mapping[str, torch.Tensor],
                             encoder_modules: Mapping[str, "torch.nn.Module"]) -> "JMIDWeights":
        """Build from the two halves of the reference checkpoint container:
        ``ckpt["ddpm"]`` (keys ``vel_predictor.net.*`` or bare net keys) and
        ``ckpt["encoder"]`` (an ``nn.ModuleDict``)  --  ``mid.py:1231-1232, 1291``."""
        out = {}
        for k, v in net_state.items():
            kk = k[len("vel_predictor.net."):] if k.startswith("vel_predictor.net.") else k
            out[kk] = v
        for mod_name in (NODE_HIST, EDGE_PED, EDGE_ROBOT, EDGE_INFL):
            for pn, p in encoder_modules[mod_name].state_dict().items():
                out[f"{mod_name}.{pn}"] = p
        return cls(dims, out)

    # ------------------------------------------------------------------ neutral file
    def save(self, path: str) -> None:
        """Neutral flat file (``.npz``); no pickled modules, loadable without the reference tree."""
        meta = np.array([self.dims.ctx_dim, self.dims.tf_layer, self.dims.nhead], dtype=np.int64)
        np.savez(path, __dims__=meta, **{k.replace("/", "|"): v.numpy() for k, v in self.tensors.items()})

    @classmethod
    def load(cls, path: str) -> "JMIDWeights":
        z = np.load(path)
        c, l, h = (int(x) for x in z["__dims__"])
        dims = NetDims(ctx_dim=c, tf_layer=l, nhead=h)
        return cls(dims, {k.replace("|", "/"): torch.from_numpy(z[k]) for k in z.files if k != "__dims__"})

    # ------------------------------------------------------------------ misc
    def checksum(self) -> str:
        """SHA-256 over names and values; computed once per object (tensors are treated as immutable afterwards;
        ``invalidate_checksum()`` after editing them in place)."""
        import hashlib

        cached = getattr(self, "_checksum", None)
        if cached is not None:
            return cached
        hsh = hashlib.sha256()
        for k, v in self.tensors.items():
            hsh.update(k.encode())
            hsh.update(v.numpy().tobytes())
        self._checksum = hsh.hexdigest()
        return self._checksum

    def invalidate_checksum(self) -> None:
        self._checksum = None

    def net_state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: v for k, v in self.tensors.items() if "/" not in k}

    def encoder_state_dicts(self) -> Dict[str, Dict[str, torch.Tensor]]:
        out: Dict[str, Dict[str, torch.Tensor]] = {}
        for k, v in self.tensors.items():
            if "/" in k:
                for mod in (NODE_HIST, EDGE_PED, EDGE_ROBOT, EDGE_INFL):
                    if k.startswith(mod + "."):
                        out.setdefault(mod, {})[k[len(mod) + 1:]] = v
        return out

    def __getitem__(self, k: str) -> torch.Tensor:
        return self.tensors[k]
