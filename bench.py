#!/usr/bin/env python
"""Headline benchmark: UNet denoise steps/sec at flow tensor (16, 2, 436->440, 1024) bf16
(BASELINE.json configs[1]).

One step = one DDPM reverse step over the batch: `model_predictions` (one UNet forward on the
HIP engine) + the fused posterior update (denoising_diffusion.py:676-698).  Inputs are synthetic
(seed 0) and resident in HBM before the timed region.  N > 1: one process per GPU, independent
chains (weak scaling, no data-path collective -- sampling shards by sample).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
"""
import argparse
import glob
import json
import os
import sys
import time

import torch

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
UNET_GFLOP_PER_SAMPLE = 1635.3   # reference module on meta, ch=5 @ 440x1024


def cpu_baseline(H, W, batch):
    """The oracle (CPU restatement of the reference UNet: same torch CPU ops, fp32) timed on this
    box's host cores on ONE sample of the workload; a batch-16 step is 16x that work."""
    from oracle import unet_ref as R
    torch.manual_seed(0)
    # the GPU box gives a one-GPU job a share of 16 host cores (cpu_count() reports the whole machine)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = int(os.environ.get("OFD_CPU_THREADS", min(avail, 16)))
    torch.set_num_threads(cores)
    P = R.closed_form_params(R.unet_param_shapes(64, 5, 2))
    x = torch.randn(1, 2, H, W)
    cond = torch.rand(1, 3, H, W) * 2 - 1
    t = torch.tensor([999])
    with torch.no_grad():
        R.unet_forward(P, x[:, :, :64, :64], cond[:, :, :64, :64], t, mode="fp32")   # warm-up of the op set
        reps = int(os.environ.get("OFD_CPU_REPS", 3))       # >= 3 timed forwards; ~20 s of CPU work in all
        t0 = time.time()
        for _ in range(reps):
            R.unet_forward(P, x, cond, t, mode="fp32")
        dt = (time.time() - t0) / reps
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": 1.0 / (dt * batch), "unit": "denoise_steps/s", "cores": torch.get_num_threads(), "cpu_model": model, "kind": "port",
            "sample": f"1 of {batch} samples: fp32 UNet forward at 1x5x{H}x{W}, mean of {reps} runs = {dt:.1f} s, scaled x{batch}"}


TRAIN_TFLOP_PER_SAMPLE = 4.8917       # fwd + bwd = 2.99 x fwd (ch=5 @ 440x1024)


def train_leg(args, dev, rank, world, H, W):
    """SURVEY 8d 'report also train steps/sec' (configs C2 / C4): FlowDiffuser.training_step (HIP training
    forward + backward) + fused Adam with clipping, per-GPU batch as the denoise leg; for world > 1 the
    gradients are averaged with the bucketed RCCL all-reduce overlapped with the backward."""
    from opticalflowdiffusion_amd import FlowDiffuser, parallel
    B = args.batch
    torch.manual_seed(0)
    fd = FlowDiffuser(dict(target="flow", image_size=[H, W], timesteps=1000, flow_max=20, zero_init=False, lr=1e-4,
                           weight_decay=0.0, clip=100.0)).to(dev)
    fd.log_dict = lambda *a, **k: None
    if world > 1:
        parallel.broadcast_parameters(fd)
        parallel.attach_grad_sync(fd)
    opt = fd.configure_optimizers()
    g = torch.Generator(device=dev).manual_seed(parallel.rank_seed(7, rank))
    img = torch.rand(B, 3, H, W, device=dev, generator=g)
    flow = torch.nn.functional.avg_pool2d(torch.clamp(torch.randn(B, 2, H, W, device=dev, generator=g) * 8, -20, 20), 9, 1, 4)

    def step(i):
        loss = fd.training_step((img, img, flow), i)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for i in range(args.train_warmup):
        step(i)
    parallel.barrier(dev)
    t0 = time.perf_counter()
    for i in range(args.train_steps):
        loss = step(i)
    parallel.barrier(dev)
    dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    sps = args.train_steps / dt
    return {"metric": "flow_diffuser train steps/sec (fwd + bwd + Adam)", "value": sps, "unit": "train_steps/s", "ms_per_step": 1e3 / sps,
            "samples_per_s": sps * B * world, "batch_per_gpu": B, "global_batch": B * world, "steps": args.train_steps,
            "warmup": args.train_warmup, "scaling": "weak", "grad_sync": parallel.describe_grad_sync(world),
            "mfma_frac_of_peak": sps * B * TRAIN_TFLOP_PER_SAMPLE * (H * W) / (440 * 1024) / PEAK_BF16_TFLOPS,
            "loss": float(loss.detach()), "max_mem_GiB": torch.cuda.max_memory_allocated(dev) / 2 ** 30}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--height", type=int, default=436)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="no per-kernel HIP events in the timed region")
    ap.add_argument("--dump-launches", default=None, help="CSV path: one row per kernel launch of the timed region")
    ap.add_argument("--train-steps", type=int, default=3, help="timed training steps of the extra `train` object (0: skip)")
    ap.add_argument("--train-warmup", type=int, default=1)
    ap.add_argument("--with-train", action="store_true", help="(accepted for compatibility: the training leg runs for every N)")
    ap.add_argument("--train-timeout", type=float, default=300.0, help="watchdog of the training leg, seconds")
    args = ap.parse_args()

    from opticalflowdiffusion_amd import parallel
    rank, local_rank, world = parallel.env_rank_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if os.environ.get("OFD_FORCE_DEVICE") is not None:       # rehearsal: several ranks on one GPU (with OFD_DIST_BACKEND=gloo)
        local_rank = int(os.environ["OFD_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        parallel.init("nccl", dev)

    from opticalflowdiffusion_amd import Unet, ConditionalDiffusion

    B = args.batch
    H = (args.height + 7) // 8 * 8        # 436 -> 440: three 2x down-samplings
    W = (args.width + 7) // 8 * 8
    torch.manual_seed(0)
    g = torch.Generator(device="cpu").manual_seed(parallel.rank_seed(1000, rank))
    unet = Unet(64, channels=5, out_dim=2, precision="bf16").to(dev)
    diff = ConditionalDiffusion(unet, (H, W), objective="pred_x0", channels=2, auto_normalize=False,
                                noise_space="image", timesteps=1000, min_snr_loss_weight=True).to(dev)
    cond = (torch.rand(B, 3, H, W, generator=g) * 2 - 1).to(dev)
    img = torch.randn(B, 2, H, W, generator=g).to(dev)
    noises = [torch.randn(B, 2, H, W, device=dev) for _ in range(2)]

    def step(img, t, i):
        out, _, _ = diff.p_sample(img, t, None, external_cond=cond, noise=noises[i & 1])
        return out

    def sync():
        torch.cuda.synchronize()
        parallel.barrier(dev)

    T = 999
    with torch.no_grad():
        for i in range(args.warmup):
            img = step(img, T - i, i)
        unet.set_profiling(not args.no_profile, args.dump_launches)
        unet.profile(reset=True)
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            img = step(img, T - args.warmup - i, i)
        sync()
        elapsed = time.perf_counter() - t0
    prof = unet.profile() if not args.no_profile else {}
    unet.set_profiling(False)
    assert torch.isfinite(img).all(), "non-finite samples"

    elapsed = parallel.max_over_ranks(elapsed, dev)

    # ---- the JSON line is assembled BEFORE the extra training leg; a watchdog guarantees it is printed exactly once even
    # if that leg hangs (e.g. a collective that never completes on some node): the headline metric must survive it
    line = None
    if rank == 0:
        steps_per_s = parallel.whole_job_rate(args.steps, world, elapsed)
        line = {
            "metric": f"UNet denoise steps/sec @ 2x{args.height}x{args.width} flow, bs={B}",
            "value": steps_per_s, "unit": "denoise_steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: flow_diffuser DDPM denoise step (UNet fwd + posterior update), "
                                   f"x (B,2,{args.height},{args.width}) padded to {H}x{W}, cond (B,3,H,W), T=1000, target=flow",
                       "batch_per_gpu": B, "global_batch": B * world, "height": H, "width": W, "parallelism": f"replicas x{world}"},
            "unet_mfma_frac_of_peak": steps_per_s / world * B * UNET_GFLOP_PER_SAMPLE * (H * W) / (440 * 1024) / 1e3 / PEAK_BF16_TFLOPS,
        }
        if prof:
            name = "conv3x3_wp_kernel<4,1>"            # dominant kernel: 25 of the 43 3x3 launches, largest total time
            k = prof[name]
            per_launch_ms = k["ms"] / max(k["launches"], 1)
            achieved = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
            # HBM bytes per launch: from separate rocprofv3 --pmc passes of THIS command (tools/pmc.sh), which cannot run inside the
            # timed region; emitted only when the committed summary was measured at this run's shape, and tagged with its file
            traffic, traffic_source = None, None
            here = os.path.dirname(os.path.abspath(__file__))
            for pmc in sorted(glob.glob(os.path.join(here, "profiles", "r*_pmc_summary.json")), reverse=True):
                d = json.load(open(pmc))
                es = [v for k, v in d["kernels"].items() if "conv3x3_wp_kernel<4, 1" in k]     # prologue on / off instantiations
                if es and d.get("shape") == [B, H, W]:
                    n = sum(v["launches"] for v in es)
                    traffic = sum((v["fetch_MB_per_launch"] + v["write_MB_per_launch"]) * v["launches"] for v in es) / n * 1e6
                    traffic_source = os.path.relpath(pmc, here)
                    break
            line["roofline"] = {"kernel": name + " (ofd::wp::conv3x3_wp_kernel<4, 1, prologue on|off>: 3x3 implicit GEMM, 128-channel blocks)", "bound": "mfma", "achieved": achieved,
                                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic,
                                "traffic_source": traffic_source,
                                "algorithmic_flops_per_launch": k["flops"] / max(k["launches"], 1), "avg_launch_ms": per_launch_ms,
                                "launches": k["launches"]}
            c3 = [prof[n] for n in ("conv3x3_wp_kernel<4,1>", "conv3x3_wp_kernel<2,2>", "conv3x3_c64_pingpong_kernel") if n in prof]
            line["conv3x3_all_tflops"] = sum(v["flops"] for v in c3) / (sum(v["ms"] for v in c3) * 1e-3) / 1e12
            line["kernel_ms_per_step"] = {n: v["ms"] / args.steps for n, v in prof.items()}
            line["hip_events_in_timed_region"] = True      # per-kernel events cost the headline number a little; --no-profile drops them

    import threading
    emitted, lock = threading.Event(), threading.Lock()

    def emit(train, hung):
        with lock:
            if not emitted.is_set():
                emitted.set()
                if rank == 0:
                    if train is not None:
                        line["train"] = train
                    print(json.dumps(line), flush=True)
        if hung:
            os._exit(3)                      # the headline line is out; a hung leg must not read as a clean run

    # GPU legs first (denoise above, training next), the CPU baseline last: the card is busy from the start of the command
    train = None
    if args.train_steps > 0:
        del img, noises
        unet._ws = None                      # hand the inference workspace back before the training one is sized
        torch.cuda.empty_cache()
        dog = threading.Timer(args.train_timeout, emit, args=({"error": f"training leg did not finish within {args.train_timeout} s"}, True))
        dog.daemon = True
        dog.start()
        try:
            train = train_leg(args, dev, rank, world, H, W)
        except Exception as e:
            train = {"error": f"{type(e).__name__}: {e}"}
        dog.cancel()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(H, W, B)
    emit(train, False)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
