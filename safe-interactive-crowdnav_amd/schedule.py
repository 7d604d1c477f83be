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
    sigmas_inflex: np.ndarray = None   # [num_steps+1] f32 (DDPM, flexibility 0; diffusion.py:41-47)
    sigmas_flex: np.ndarray = None     # [num_steps+1] f32 = sqrt(betas) (flexibility 1; diffusion.py:41)

    @classmethod
    def linear(cls, num_steps: int = 100, beta_1: float = 1e-4, beta_T: float = 5e-2) -> "VarianceSchedule":
        betas = torch.linspace(beta_1, beta_T, steps=num_steps)
        betas = torch.cat([torch.zeros([1]), betas], dim=0)
        alphas = 1 - betas
        log_alphas = torch.log(alphas)
        for i in range(1, log_alphas.size(0)):
            log_alphas[i] += log_alphas[i - 1]
        alpha_bars = log_alphas.exp()
        sigmas_flex = torch.sqrt(betas)
        sigmas_inflex = torch.zeros_like(betas)
        for i in range(1, betas.size(0)):
            sigmas_inflex[i] = ((1 - alpha_bars[i - 1]) / (1 - alpha_bars[i])) * betas[i]
        sigmas_inflex = torch.sqrt(sigmas_inflex)
        return cls(betas.numpy().copy(), alphas.numpy().copy(), alpha_bars.numpy().copy(), num_steps,
                   sigmas_inflex.numpy().copy(), sigmas_flex.numpy().copy())


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


@dataclass(frozen=True)
class DDPMStep:
    t: int
    beta: np.float32
    c0: np.float32       # 1/sqrt(alpha_t)
    c1: np.float32       # (1 - alpha_t)/sqrt(1 - abar_t)
    sigma: np.float32    # get_sigmas(t, flexibility) (diffusion.py:59-64)
    noise: bool          # z ~ N(0,1) is used (t > 1), else zeros     (diffusion.py:509)


def ddpm_steps(sched: VarianceSchedule, step: int, flexibility: float = 0.0) -> List[DDPMStep]:
    """x_next = c0*(x - c1*e) + sigma*z  (``sampling="ddpm"``, diffusion.py:509-522), same step enumeration as DDIM;
    sigma = sigmas_flex[t]*flexibility + sigmas_inflex[t]*(1 - flexibility) (``get_sigmas``, diffusion.py:59-64)."""
    if not 0.0 <= flexibility <= 1.0:
        raise ValueError("flexibility must be in [0, 1]")
    if step <= 0:
        raise ValueError("step must be positive")
    stride = int(100 / step)
    if stride <= 0 or sched.num_steps % stride != 0:
        raise ValueError(f"step={step}: stride {stride} does not divide num_steps={sched.num_steps}")
    al, ab = torch.from_numpy(sched.alphas), torch.from_numpy(sched.alpha_bars)
    sg = torch.from_numpy(sched.sigmas_flex) * flexibility + torch.from_numpy(sched.sigmas_inflex) * (1 - flexibility)
    out = []
    for t in range(sched.num_steps, 0, -stride):
        out.append(DDPMStep(t=t, beta=np.float32(sched.betas[t]), c0=np.float32((1.0 / torch.sqrt(al[t])).item()),
                            c1=np.float32(((1 - al[t]) / torch.sqrt(1 - ab[t])).item()), sigma=np.float32(sg[t].item()),
                            noise=t > 1))
    return out
