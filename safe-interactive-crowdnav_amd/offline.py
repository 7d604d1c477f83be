"""Per-sample sampling used by the reference's offline evaluation, on the HIP engine.

Mirrors ``DiffusionTraj.sample`` (``sicnav_diffusion/JMID/MID/models/diffusion.py:544-613``) and the tail of
``AutoEncoder.generate`` (``MID/models/autoencoder.py:50-103``): unlike ``sample_sicnav_inference`` (the MPC-step path,
``engine.JmidEngine.denoise`` with K samples in one batch), every sample is denoised on its own with the context
rows as the batch, so for JMID a sample is one attention sequence of B*T tokens.  On the engine that is
``E = sample`` independent "episodes" with K = 1 that share the context - one batched launch instead of the
reference's Python loop over samples.

RNG contract: the global torch CPU generator is consumed exactly as the reference does - per sample ``x_T``
(``bestof``; zeros otherwise, no draw), then one ``randn_like`` per step for t > 1 (``z``; drawn for "ddim" as well,
where it is unused).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from .engine import JmidEngine


def sample(engine: JmidEngine, num_points: int, context, sample: int, bestof: bool, point_dim: int = 2,
           flexibility: float = 0.0, ret_traj: bool = False, sampling: str = "ddpm", step: int = 100,
           precision: str = "f32", _integrate=None) -> Tuple[np.ndarray, int, int, int, int]:
    """-> (vel [sample, B, num_points, 2] float32, number_of_steps, 0, 0, 0)   (diffusion.py:603-613).
    ``_integrate`` = (p0 [B, 2], dt): used by ``generate`` - the same call also integrates on the device (integrate_kernel) and
    the first element of the result is the positions instead."""
    if point_dim != 2:
        raise ValueError("point_dim must be 2")
    if ret_traj:
        raise NotImplementedError("ret_traj=True (every intermediate x_t) is a training-time debugging aid")
    if sampling not in ("ddpm", "ddim"):
        raise ValueError("sampling must be 'ddpm' or 'ddim'")      # the reference drops into pdb here (:595)
    ctx = torch.as_tensor(np.asarray(context, dtype=np.float32))
    B = int(ctx.shape[0])
    engine.set_step(step, sampling, flexibility)
    n_steps, stride = engine.n_steps, int(100 / step)
    x_T = torch.zeros([sample, B, num_points, 2])
    z = torch.zeros([n_steps, sample, B, num_points, 2])
    for i in range(sample):
        if bestof:
            x_T[i] = torch.randn([B, num_points, point_dim])
        for k, t in enumerate(range(engine.schedule.num_steps, 0, -stride)):
            if t > 1:
                z[k, i] = torch.randn([B, num_points, point_dim])
    ctx_e = ctx.unsqueeze(0).expand(sample, B, ctx.shape[1]).contiguous().numpy()
    number_of_steps = sample * (engine.schedule.num_steps // stride + 1)
    if _integrate is not None:      # every sample is an "episode" of B agents with K = 1: p0 repeats per sample
        p0 = np.ascontiguousarray(np.broadcast_to(np.asarray(_integrate[0], dtype=np.float32)[None], (sample, B, 2)))
        _, pos = engine.denoise(x_T.numpy(), ctx_e, p0, dt=float(_integrate[1]), precision=precision, want_vel=False,
                                z=z.numpy() if sampling == "ddpm" else None)
        return pos.reshape(sample, B, num_points, 2), number_of_steps, 0, 0, 0
    vel, _ = engine.denoise(x_T.numpy(), ctx_e, None, precision=precision, want_pos=False,
                            z=z.numpy() if sampling == "ddpm" else None)
    return vel.reshape(sample, B, num_points, 2), number_of_steps, 0, 0, 0


def generate(engine: JmidEngine, context, p0, dt: float, num_points: int, sample_n: int, bestof: bool,
             flexibility: float = 0.0, sampling: str = "ddpm", step: int = 100, precision: str = "f32"):
    """Tail of ``AutoEncoder.generate``: ``sample`` + ``SingleIntegrator.integrate_samples``
    (``single_integrator.py:290-321``): pos = cumsum(vel, T) * dt + p0[b].  ``context`` [B, ctx] is the encoder
    output (``JmidEngine.encode``), ``p0`` [B, 2] the current positions.  -> (pos [sample, B, T, 2], steps, 0, 0, 0)."""
    pos, nsteps, a, b, c = sample(engine, num_points, context, sample_n, bestof, flexibility=flexibility,
                                  sampling=sampling, step=step, precision=precision, _integrate=(p0, dt))
    return np.asarray(pos, dtype=np.float32), nsteps, a, b, c
