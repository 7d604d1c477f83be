"""Diffusion noise schedule and the DDIM step table (host side).

Follows ``VarianceSchedule`` (``sicnav_diffusion/JMID/MID/models/diffusion.py:12-64``) as
built by ``MID._build_model`` (``MID/mid.py:1281-1283``: ``num_steps=100, beta_1=1e-4,
beta_T=5e-2, mode="linear"``), including its fp32 sequential accumulation of
``log(alpha)`` (``diffusion.py:36-39``), and the step enumeration of
``DiffusionTraj.sample_sicnav_inference`` (``diffusion.py:507-528``).

torch is used only for bit-identical fp32 ``linspace``/``log``/``exp`` on the host.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np
import torch


@dataclass(frozen=True)
class VarianceSchedule:
    betas: np.ndarray        # [num_steps+1] f32, betas[0] = 0 (padding)
    alphas: np.ndarray       # [num_steps+1] f32
    alpha_bars: np.ndarray   # [num_steps+1] f32
    num_steps: int

    @classmethod
    def linear(cls, num_steps: int = 100, beta_1: float = 1e-4, beta_T: float = 5e-2) -> "VarianceSchedule":
        betas = torch.linspace(beta_1, beta_T, steps=num_steps)
        betas = torch.cat([torch.zeros([1]), betas], dim=0)
        alphas = 1 - betas
        log_alphas = torch.log(alphas)
        for i in range(1, log_alphas.size(0)):
            log_alphas[i] += log_alphas[i - 1]
        alpha_bars = log_alphas.exp()
        return cls(betas.numpy().copy(), alphas.numpy().copy(), alpha_bars.numpy().copy(), num_steps)


@dataclass(frozen=True)
class DDIMStep:
    t: int
    beta: np.float32
    # x0 = (x - e*c_e)/c_x ; x_next = n_x*x0 + n_e*e
    c_e: np.float32    # sqrt(1 - abar_t)
    c_x: np.float32    # sqrt(abar_t)
    n_x: np.float32    # sqrt(abar_{t-stride})
    n_e: np.float32    # sqrt(1 - abar_{t-stride})


def ddim_steps(sched: VarianceSchedule, step: int) -> List[DDIMStep]:
    """Steps of the reverse loop for ``step`` (the reference's ``step_size`` config key).

    ``stride = int(100 / step)`` and ``t = num_steps, num_steps-stride, ... > 0``
    (``diffusion.py:507-508``).  Configs whose stride does not divide ``num_steps``
    would index ``alpha_bars[t - stride]`` with a negative index in the reference
    (Python wrap-around); they are rejected here.
    """
    if step <= 0:
        raise ValueError("step must be positive")
    stride = int(100 / step)
    if stride <= 0 or sched.num_steps % stride != 0:
        raise ValueError(
            f"step={step} gives stride {stride}, which does not divide num_steps={sched.num_steps}; "
            "the reference would wrap alpha_bars[t-stride] around (undefined schedule)"
        )
    out = []
    ab = torch.from_numpy(sched.alpha_bars)
    for t in range(sched.num_steps, 0, -stride):
        a_t, a_n = ab[t], ab[t - stride]
        out.append(
            DDIMStep(
                t=t,
                beta=np.float32(sched.betas[t]),
                c_e=np.float32((1 - a_t).sqrt().item()),
                c_x=np.float32(a_t.sqrt().item()),
                n_x=np.float32(a_n.sqrt().item()),
                n_e=np.float32((1 - a_n).sqrt().item()),
            )
        )
    return out
