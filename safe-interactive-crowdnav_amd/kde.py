"""Joint-KDE ranking of the sampled futures (host side, tiny).

Mirror of ``get_most_likely_samples`` (``sicnav_diffusion/JMID/mid_sim_wrapper.py:14-169``), which always takes its
*joint* branch for this predictor (``hasattr(mid_model, "cfg")`` is False, ``:20-21``): per horizon step a Gaussian
KDE over the K samples in R^{2A} with bandwidth exp(linspace(ln .01, ln .1, H)), log-likelihood of every sample
under it, normalised over samples, summed over the horizon; the k most likely samples are kept (ascending order,
as ``argsort(...)[-k:]``) and their log-weights renormalised.  The arithmetic is fp32 on small [H, 2A, 2A] matrices;
torch's CPU linalg is used as the container for it (same routine choice as the reference: inverse -> cholesky_ex
-> inv).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np
import torch


def most_likely_samples(forecasts: np.ndarray, num_ret: int) -> Tuple[np.ndarray, np.ndarray]:
    """forecasts [K, A, H, 2] -> (kept [A, k, H, 2] float32, log-weights [A, k] float32)."""
    f = torch.as_tensor(np.ascontiguousarray(forecasts), dtype=torch.float32)
    K, A, H, _ = f.shape
    d = 2 * A
    pts = f.permute(2, 0, 1, 3).reshape(H, K, d)                                  # horiz, samples, (humans xy)
    bw = torch.exp(torch.linspace(math.log(0.01), math.log(0.1), steps=H))
    n = torch.tensor(float(K), dtype=torch.float32)
    centered = pts - pts.mean(dim=1, keepdim=True)
    cov = torch.bmm(centered.transpose(1, 2), centered) / (n - 1)
    prec = bw[:, None, None] ** -2 * cov + torch.eye(d).expand_as(cov) * 1e-6
    L = torch.linalg.cholesky_ex(torch.inverse(prec))[0]
    diffs = pts.unsqueeze(2) - pts.unsqueeze(1)                                   # [H, K, K, d]
    diffs = torch.matmul(diffs, torch.linalg.inv(L).unsqueeze(1)) / bw[:, None, None, None]
    log_exp = -0.5 * torch.norm(diffs, p=2, dim=-1) ** 2
    log_det = 2 * torch.sum(torch.log(torch.diagonal(L, dim1=-2, dim2=-1)), dim=-1)
    Z = 0.5 * d * torch.log(2 * torch.tensor(math.pi)) + 0.5 * log_det.unsqueeze(-1) + torch.log(n)
    ll = torch.logsumexp(log_exp - Z.unsqueeze(-1), dim=-1)                       # [H, K]
    ll = ll - torch.logsumexp(ll, dim=1, keepdim=True)
    total = ll.sum(dim=0)                                                          # [K]
    keep = torch.argsort(total, dim=-1)[-num_ret:]
    logw = total[keep]
    logw = logw - torch.logsumexp(logw, dim=-1, keepdim=True)
    kept = f[keep].permute(1, 0, 2, 3).contiguous()
    return kept.numpy(), logw.unsqueeze(0).expand(A, num_ret).contiguous().numpy()
