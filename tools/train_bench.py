"""Training-step timing of the FlowDiffuser path (SURVEY config C2 / C4): forward + backward + fused Adam on
synthetic Sintel-shaped batches, one process per GPU, gradients averaged with the bucketed all-reduce.

    python tools/train_bench.py --batch 16 --height 440 --width 1024 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py ...
Prints one JSON line on rank 0 (train steps/s over the whole job; samples/s; per-class kernel time when --profile)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflowdiffusion_amd import FlowDiffuser, parallel as P   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch (weak scaling, as DDP)")
    ap.add_argument("--height", type=int, default=440)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--target", default="flow", choices=["flow", "joint", "target"],
                    help="flow_diffuser.yaml `target` (the shipped default is joint: UnetWithWarp + pyramid loss; needs H, W % 16 == 0)")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--dump", default=None)
    a = ap.parse_args()
    rank, local_rank, world = P.env_rank_world()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    P.init(device=dev)
    torch.manual_seed(0)
    fd = FlowDiffuser(dict(target=a.target, image_size=[a.height, a.width], timesteps=1000, flow_max=20, zero_init=False,
                           lr=1e-4, weight_decay=0.0, clip=100.0)).to(dev)
    fd.log_dict = lambda *x, **k: None
    P.broadcast_parameters(fd)
    if world > 1:
        P.attach_grad_sync(fd)
    opt = fd.configure_optimizers()
    g = torch.Generator(device=dev).manual_seed(P.rank_seed(0, rank))
    B, H, W = a.batch, a.height, a.width
    img = torch.rand(B, 3, H, W, device=dev, generator=g)
    flow = torch.clamp(torch.randn(B, 2, H, W, device=dev, generator=g) * 8, -20, 20)
    flow = torch.nn.functional.avg_pool2d(flow, 9, 1, 4)

    def step(i):
        loss = fd.training_step((img, img, flow), i)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for i in range(a.warmup):
        step(i)
    if a.profile:
        fd.unet.set_profiling(True, a.dump)
        fd.unet.profile(reset=True)
    P.barrier(dev)
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(i)
    P.barrier(dev)
    dt = P.max_over_ranks(time.perf_counter() - t0, dev)
    res = {"metric": "FlowDiffuser train steps/sec", "value": P.whole_job_rate(a.steps, 1, dt), "unit": "steps/s", "n_gpus": world,
           "samples_per_s": P.whole_job_rate(a.steps * B, world, dt), "ms_per_step": 1e3 * dt / a.steps, "steps": a.steps, "warmup": a.warmup,
           "scaling": "weak", "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"train target={a.target} B={B}/GPU {H}x{W} T=1000 Adam", "global_batch": B * world},
           "loss": float(loss), "max_mem_GiB": torch.cuda.max_memory_allocated(dev) / 2 ** 30}
    if a.profile:
        prof = fd.unet.profile()
        res["kernel_ms_per_step"] = {k: round(v["ms"] / a.steps, 3) for k, v in prof.items() if v["launches"]}
        res["kernel_tflops"] = {k: round(v["flops"] / v["ms"] / 1e9, 1) for k, v in prof.items() if v["launches"] and v["flops"] > 0 and v["ms"] > 0}
    if rank == 0:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
