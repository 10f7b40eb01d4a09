"""Multi-GPU helpers of the FlowDiffuser path: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Sampling (the benchmarked path) shards BY SAMPLE: every rank runs independent chains on its own
slice of the global batch, so there is no data-path collective -- only a barrier around the timed
region and a MAX-reduction of the elapsed time.  The reference's only parallelism is
Lightning DDP (experiments/exp_base.py:198): its gradient all-reduce is `BucketedAllReduce` below, fed by the
range callback of `ofd_unet_backward` and overlapped with the remaining backward launches.
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
        # OFD_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (gloo moves CUDA
        # tensors through the host; RCCL refuses two ranks on one device)
        backend = os.environ.get("OFD_DIST_BACKEND") or backend or ("nccl" if torch.cuda.is_available() else "gloo")
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


def describe_grad_sync(world, bucket_bytes=32 << 20, mechanism="hook", bucket_dtype="fp32"):
    """what actually synchronises the gradients in this process group (for benchmark records).  mechanism: "hook" = the executor's
    range callback feeding BucketedAllReduce; "torch" = torch.nn.parallel.DistributedDataParallel's reducer (TorchDDP)"""
    if world <= 1 or not (dist.is_available() and dist.is_initialized()):
        return "none"
    be = dist.get_backend()
    name = {"nccl": "RCCL (torch backend nccl)", "gloo": "gloo (host transport: rehearsal, not xGMI)"}.get(be, be)
    if mechanism == "torch":
        return f"torch DistributedDataParallel reducer over {name} (autograd hooks; no_sync on all but the last micro-batch)"
    return (f"bucketed all-reduce over {name}, {bucket_bytes >> 20} MiB buckets of {bucket_dtype}, side stream, overlapped with backward "
            "(once per backward: every micro-batch of an accumulated step)")


def whole_job_rate(units_per_rank, world, seconds):
    """aggregate throughput: units all ranks processed / time of the slowest rank"""
    return units_per_rank * world / seconds


class BucketedAllReduce:
    """Gradient averaging of data-parallel training (the reference's DDPStrategy, exp_base.py:193-206)
    for the UNet executor's FLAT gradient buffer.

    `Unet._backward` reports [begin, end) float ranges of that buffer in backward order, each as soon
    as the launches producing it are enqueued.  Ranges are coalesced (neighbouring ops are adjacent in
    the buffer) and, once `bucket_bytes` are pending, all-reduced on a side stream that waits on an
    event recorded on the compute stream -- the collective of one bucket overlaps the backward of the
    layers below it.  xGMI is point-to-point: few large messages (default 32 MiB) beat many small ones.
    `finish` flushes the tail and makes the compute stream wait for every collective.

    `bucket_dtype="bf16"` (71.5 MB instead of 142.9 MB per step on the wire): each range is cast INTO a bf16 bucket on
    the side stream, the bucket is all-reduced, and the sum is cast back into the flat fp32 buffer before the 1/world scaling
    (which stays fp32).  The gradients are rounded to 8 significant bits once before the sum and the sum once after it; fp32 is
    the default until an 8-GPU run can compare convergence.
    """

    def __init__(self, bucket_bytes=32 << 20, group=None, bucket_dtype="fp32"):
        if bucket_dtype not in ("fp32", "bf16"):
            raise ValueError(f"bucket_dtype {bucket_dtype!r}: 'fp32' or 'bf16'")
        self.bucket_dtype = bucket_dtype
        self.bucket_bytes = int(bucket_bytes)
        self.group = group
        self._comm = None
        self._pending, self._pending_floats, self._works, self._done = [], 0, [], []
        self._casts = []

    def _world(self):
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def begin(self, flat):
        self._pending, self._pending_floats, self._works, self._done = [], 0, [], []
        self._casts = []                      # bf16 buckets in flight: (begin, end, bucket)
        if flat.is_cuda and self._comm is None:
            self._comm = torch.cuda.Stream(device=flat.device)

    def on_range(self, flat, begin, end):
        if end <= begin:
            return
        p = self._pending
        if p and end == p[-1][0]:
            p[-1] = (begin, p[-1][1])
        elif p and begin == p[-1][1]:
            p[-1] = (p[-1][0], end)
        else:
            p.append((begin, end))
        self._pending_floats += end - begin
        if self._pending_floats * 4 >= self.bucket_bytes:
            self._flush(flat)

    def _flush(self, flat):
        ranges, self._pending, self._pending_floats = self._pending, [], 0
        if not ranges:
            return
        self._done.extend(ranges)
        if self._world() == 1:
            return
        bf16 = self.bucket_dtype == "bf16"
        if flat.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(flat.device))
            self._comm.wait_event(ev)
            with torch.cuda.stream(self._comm):
                for b, e in ranges:
                    buf = flat[b:e]
                    if bf16:
                        buf = flat[b:e].to(torch.bfloat16)             # cast-into-bucket on the side stream
                        self._casts.append((b, e, buf))
                    self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            for b, e in ranges:
                if bf16:
                    buf = flat[b:e].to(torch.bfloat16)
                    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                    flat[b:e].copy_(buf)
                else:
                    dist.all_reduce(flat[b:e], op=dist.ReduceOp.SUM, group=self.group)

    def finish(self, flat):
        self._flush(flat)
        world = self._world()
        if world > 1:
            if flat.is_cuda:
                with torch.cuda.stream(self._comm):
                    for w in self._works:
                        w.wait()
                    for b, e, buf in self._casts:                      # the bf16 sums back into the flat fp32 buffer
                        flat[b:e].copy_(buf)
                    for b, e in self._done:
                        flat[b:e].mul_(1.0 / world)
                torch.cuda.current_stream(flat.device).wait_stream(self._comm)
            else:
                for b, e in self._done:
                    flat[b:e].mul_(1.0 / world)
        self._works, self._casts = [], []
        return self._done


def broadcast_parameters(module, src=0, group=None):
    """rank `src`'s parameters and buffers to every rank (what DDP does at construction)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def attach_grad_sync(flow_diffuser_or_unet, bucket_bytes=32 << 20, group=None, bucket_dtype="fp32"):
    """data-parallel training: average the UNet's gradients across ranks inside every backward."""
    from .denoising_diffusion import Unet
    root = flow_diffuser_or_unet
    unets = [m for m in root.modules() if isinstance(m, Unet)] if isinstance(root, torch.nn.Module) else []
    if len(unets) != 1:        # FlowDiffuser.unet, FlowLearner.unet.model (inside UnetWithWarp), or a bare Unet
        raise ValueError(f"attach_grad_sync: expected exactly one engine Unet under the module, found {len(unets)}")
    unets[0].grad_sync = BucketedAllReduce(bucket_bytes, group, bucket_dtype)
    return unets[0].grad_sync


class TorchDDP(torch.nn.Module):
    """The plugin wrapped the way Lightning's DDPStrategy wraps it (exp_base.py:198): torch.nn.parallel.DistributedDataParallel
    around a module whose forward is `training_step`.  The engine's parameters are ordinary leaf nn.Parameters fed by a custom
    autograd Function, so DDP's reducer hooks see them like any other module's (tests/test_trainer_gpu.py)."""

    class _Step(torch.nn.Module):
        def __init__(self, plugin):
            super().__init__()
            self.plugin = plugin

        def forward(self, batch, batch_idx):
            return self.plugin.training_step(batch, batch_idx)

    def __init__(self, plugin, device=None, **ddp_kwargs):
        super().__init__()
        ids = None
        if device is not None and torch.device(device).type == "cuda":
            idx = torch.device(device).index
            ids = [torch.cuda.current_device() if idx is None else idx]      # torch.device("cuda") carries no index: DDP rejects [None]
        self.ddp = torch.nn.parallel.DistributedDataParallel(TorchDDP._Step(plugin), device_ids=ids, find_unused_parameters=False, **ddp_kwargs)

    def training_step(self, batch, batch_idx):
        return self.ddp(batch, batch_idx)

    def no_sync(self):
        """context for every micro-batch of an accumulated step but the last (DDP then all-reduces once per optimizer step)"""
        return self.ddp.no_sync()
