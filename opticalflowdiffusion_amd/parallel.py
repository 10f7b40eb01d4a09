"""Multi-GPU helpers of the FlowDiffuser path: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Sampling (the benchmarked path) shards BY SAMPLE: every rank runs independent chains on its own
slice of the global batch, so there is no data-path collective -- only a barrier around the timed
region and a MAX-reduction of the elapsed time.  The reference's only parallelism is
Lightning DDP (experiments/exp_base.py:198); its gradient all-reduce belongs to the training
path, which needs the backward kernels.
"""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, device=None):
    """init_process_group from the torchrun environment; no-op for a single process."""
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def shard_batch(global_batch, rank, world):
    """[start, stop) of this rank's samples: contiguous, sizes differ by at most one."""
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def rank_seed(base_seed, rank):
    """independent noise streams per rank (disjoint chains)"""
    return base_seed + 1000003 * rank


def barrier(device=None):
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(seconds, device=None):
    """elapsed time of the slowest rank (what the whole job took)"""
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_rate(units_per_rank, world, seconds):
    """aggregate throughput: units all ranks processed / time of the slowest rank"""
    return units_per_rank * world / seconds
