"""The only cross-GPU exchange of this path: sequences are independent (one batch of sequences per GPU, no data-path
collective), so the end-of-run throughput reduction is a MAX over elapsed seconds and a SUM over counters --
RCCL (backend "nccl") on the GPUs over xGMI, gloo in the CPU tests."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend, device=None):
    """init torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them); no-op for 1 rank."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
    dist.init_process_group(backend, **kw)
    return True


def aggregate(elapsed_s, counters, device="cpu"):
    """returns (max elapsed over ranks, element-wise sum of counters over ranks)."""
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    c = torch.tensor([float(x) for x in counters], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(x) for x in c.tolist()]


def shard_sequences(n_sequences, rank, world):
    """config 5 (8 EuRoC sequences on 8 GPUs): sequence i -> rank i % world; returns this rank's sequence ids."""
    return [i for i in range(n_sequences) if i % world == rank]
