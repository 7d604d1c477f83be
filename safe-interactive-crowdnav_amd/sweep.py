"""Multi-GPU evaluation sweep: independent episodes shard across ranks (one process per GPU), no collective on the
data path; a single gather of the per-episode metrics over RCCL (``torch.distributed`` backend "nccl" on ROCm) at
the end.  JMID attention is intra-episode, so an episode is never split.

The reference has no distributed code; this is the build's own harness for BASELINE configs 3 and 5.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Block partition: rank g owns episodes [g*total/world, (g+1)*total/world) (balanced to +-1)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_metrics(local: torch.Tensor, total: int, dst: int = 0, force: bool = False) -> Optional[np.ndarray]:
    """Gather per-episode metric rows [E_local, M] from every rank to ``dst`` in episode order.
    One collective of a few KB (latency-bound; payload is independent of the model size).  ``force`` runs the collective
    even in a one-rank group (tests of the RCCL call on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return local.detach().cpu().numpy()
    world, rank = dist.get_world_size(), dist.get_rank()
    if dist.get_backend() == "gloo" and local.is_cuda:     # host collective (CPU tests; several ranks sharing one GPU)
        local = local.detach().cpu()
    width = local.shape[1]
    cap = max(shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world))
    buf = torch.full((cap, width), float("nan"), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    gl = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gl, dst=dst)
    if rank != dst:
        return None
    rows = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        rows.append(gl[r][: hi - lo].cpu().numpy())
    return np.concatenate(rows, axis=0)
