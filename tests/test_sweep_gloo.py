"""N>1 path of the evaluation sweep on CPU: world_size-2 gloo processes shard episodes and gather metrics."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from safe_interactive_crowdnav_amd.sweep import gather_metrics, shard_range


def test_shard_range_partitions_everything():
    for total in (1, 7, 256, 4096):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(total, rank, world)
    local = torch.stack([torch.arange(lo, hi, dtype=torch.float32) * (c + 1) for c in range(4)], dim=1)
    out = gather_metrics(local, total)
    if rank == 0:
        q.put(out)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gather_metrics_world2_gloo():
    total, world = 7, 2       # ragged shards (4 + 3)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    expect = np.stack([np.arange(total, dtype=np.float32) * (c + 1) for c in range(4)], axis=1)
    np.testing.assert_array_equal(out, expect)


def test_gather_metrics_single_process():
    local = torch.arange(12, dtype=torch.float32).view(3, 4)
    np.testing.assert_array_equal(gather_metrics(local, 3), local.numpy())
