#!/usr/bin/env python
"""Headline benchmark: UNet denoise steps/sec at flow tensor (16, 2, 436->440, 1024) bf16
(BASELINE.json configs[1]).

One step = one DDPM reverse step over the batch: `model_predictions` (one UNet forward on the
HIP engine) + the per-step noise draw (DD:687) + the fused posterior update (denoising_diffusion.py:676-698).
Inputs are synthetic (seed 0) and resident in HBM before the timed region.  N > 1: one process per GPU,
independent chains (weak scaling, no data-path collective -- sampling shards by sample).

The headline loop carries no per-kernel instrumentation; a second, instrumented loop after it feeds `roofline` /
`kernel_ms_per_step`; `warp` is the flow-warp leg (splat / grid_sample GB/s); `train` the training step.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python bench.py --gpus N ...            (starts its own N ranks as a child `python -m torch.distributed.run`)
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


# ---- roofline of the instrumented loop -----------------------------------------------------------------------------------------------
# A profile class is named `<kernel family> [<layer group>]` by the library (csrc/unet.hip:prof_class_name, from the switches the process runs
# with); classes served by one kernel template share the family.  What bounds a family:
HBM_FAMILIES = ("conv1x1", "resblock_out", "layernorm_c", "gn_finalize", "misc")
VALU_FAMILIES = ("la_ctx_fused",)            # LinearAttention: transcendental / latency-bound passes with MFMA projections inside


def family_of(name):
    return name.split(" [")[0]


def bound_of(name):
    fam = family_of(name)
    if fam.startswith(HBM_FAMILIES):
        return "hbm"
    if fam.startswith(VALU_FAMILIES):
        return "valu"
    return "mfma"


def pmc_traffic(family, B, H, W):
    """HBM bytes per launch of a kernel family from the newest committed rocprofv3 --pmc summary measured at this shape (separate passes of
    this command, tools/pmc.sh: counters cannot be collected inside the timed region); (None, None) when there is none"""
    here = os.path.dirname(os.path.abspath(__file__))
    key = family.split("|")[0].replace(" ", "").rstrip(">")          # "conv3x3_wp_kernel<2,2"
    for pmc in sorted(glob.glob(os.path.join(here, "profiles", "r*_pmc_summary.json")), reverse=True):
        d = json.load(open(pmc))
        es = [v for k, v in d["kernels"].items() if key in k.replace(" ", "")]
        if es and d.get("shape") == [B, H, W]:
            n = sum(v["launches"] for v in es)
            return sum((v["fetch_MB_per_launch"] + v["write_MB_per_launch"]) * v["launches"] for v in es) / n * 1e6, os.path.relpath(pmc, here)
    return None, None


def roofline_objects(prof, prof_steps, prof_elapsed, B, H, W):
    """`roofline` (the MFMA kernel family with the largest measured time per step -- picked from the measurement, not by name),
    `roofline_classes` (every profile class that ran: ms, work, achieved rate against the roof that bounds it) and `kernel_ms_per_step`
    (per kernel family, the classes of a family as sub-fields)."""
    ran = {n: v for n, v in prof.items() if v["launches"] > 0}
    classes, fams = [], {}
    for n, v in ran.items():
        ms, bound = v["ms"] / prof_steps, bound_of(n)
        row = {"class": n, "bound": bound, "ms_per_step": ms, "launches_per_step": v["launches"] / prof_steps}
        if v["flops"] > 0:
            row["tflop_per_step"] = v["flops"] / prof_steps / 1e12
            row["achieved_TFLOPs"] = v["flops"] / (v["ms"] * 1e-3) / 1e12
            row["frac_of_mfma_peak"] = row["achieved_TFLOPs"] / PEAK_BF16_TFLOPS
        if bound == "hbm" and v["bytes"] > 0:
            row["achieved_GBps"] = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            row["frac_of_hbm_peak"] = row["achieved_GBps"] / PEAK_HBM_GBS
        classes.append(row)
        f = fams.setdefault(family_of(n), {"ms": 0.0, "flops": 0.0, "launches": 0, "bound": bound, "classes": {}})
        f["ms"] += v["ms"]; f["flops"] += v["flops"]; f["launches"] += v["launches"]
        f["classes"][n] = ms
    classes.sort(key=lambda r: -r["ms_per_step"])
    out = {"roofline_classes": classes,
           "kernel_ms_per_step": {fam: ({"total": f["ms"] / prof_steps, **f["classes"]} if len(f["classes"]) > 1 else f["ms"] / prof_steps)
                                  for fam, f in sorted(fams.items(), key=lambda kv: -kv[1]["ms"])}}
    mf = {fam: f for fam, f in fams.items() if f["bound"] == "mfma" and f["flops"] > 0}
    if mf:
        fam, f = max(mf.items(), key=lambda kv: kv[1]["ms"])
        achieved = f["flops"] / (f["ms"] * 1e-3) / 1e12
        traffic, traffic_source = pmc_traffic(fam, B, H, W)
        out["roofline"] = {"kernel": fam + " (the MFMA kernel family with the largest time per step in this run; its classes: " + "; ".join(f["classes"]) + ")",
                           "bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                           "traffic": traffic, "traffic_source": traffic_source,
                           "algorithmic_flops_per_launch": f["flops"] / f["launches"], "avg_launch_ms": f["ms"] / f["launches"],
                           "launches": f["launches"], "ms_per_step": f["ms"] / prof_steps,
                           "measured_in": f"instrumented loop of {prof_steps} steps after the headline loop ({1e3 * prof_elapsed / prof_steps:.2f} ms per step with the events on, "
                                          "one stream: the headline loop runs the two half-batches on two streams, where a kernel's duration includes what it shares the chip with)"}
        c3 = [f for f in mf.values() if any("[3x3" in c for c in f["classes"])]
        if c3:
            out["conv3x3_all_tflops"] = sum(f["flops"] for f in c3) / (sum(f["ms"] for f in c3) * 1e-3) / 1e12
    return out


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
        parallel.attach_grad_sync(fd, bucket_dtype=args.grad_bucket_dtype)
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
    return {"metric": "flow_diffuser train steps/sec (fwd + bwd + Adam)",
            "includes": "FlowDiffuser.training_step = GPU augmentation (ofd_augment, cfg.augment default true as FD:219) + preprocess + q_sample + "
                        "UNet training forward + loss; backward; global-norm clip + fused Adam",
            "value": sps, "unit": "train_steps/s", "ms_per_step": 1e3 / sps,
            "samples_per_s": sps * B * world, "batch_per_gpu": B, "global_batch": B * world, "steps": args.train_steps,
            "warmup": args.train_warmup, "scaling": "weak", "grad_sync": parallel.describe_grad_sync(world, bucket_dtype=args.grad_bucket_dtype),
            "mfma_frac_of_peak": sps * B * TRAIN_TFLOP_PER_SAMPLE * (H * W) / (440 * 1024) / PEAK_BF16_TFLOPS,
            "loss": float(loss.detach()), "max_mem_GiB": torch.cuda.max_memory_allocated(dev) / 2 ** 30}


WARP_RADIUS = 24               # window half-width of the splat / grid-warp tile kernels: > the +-20 px clamp of the synthetic flow


def workload_name(B, H, W, h_arg, w_arg):
    """config.workload from the ACTUAL arguments (BASELINE.json configs[] named where the shape is one of them)"""
    base = (f"flow_diffuser DDPM denoise step (UNet fwd + noise draw + posterior update), x ({B},2,{h_arg},{w_arg}) padded to {H}x{W}, "
            f"cond ({B},3,{H},{W}), T=1000, target=flow")
    if (B, H, W) == (16, 440, 1024):
        return "BASELINE configs[1]: " + base
    if (H, W) == (1080, 1920):
        return f"BASELINE configs[4] (1080p, bs=8 over 8 GPUs) per-GPU share at B={B}: " + base
    return "custom shape: " + base


def smooth_flow(B, H, W, dev, gen):
    """SURVEY 8d synthetic flow: N(0, 8^2) px, 9x9 box filter, clamped to +-20 px"""
    f = torch.randn(B, 2, H, W, device=dev, generator=gen) * 8 * 9
    return torch.nn.functional.avg_pool2d(f, 9, 1, 4).clamp(-20, 20).contiguous()


def warp_leg(B, H, W, dev, reps=20):
    """SURVEY 8d 'splat GB/s separately': the two flow-warp kernels of the path at the bench shape, inputs resident, HIP events on
    the launch stream (torch's current stream: the library launches there), against the algorithmic bytes of SURVEY 8d:
    forward splat (SS:352-423, C = 3 + 1 weight channel) 40 B/px; grid_sample warp with mask (WP:95-119, C = 3) 44 B/px."""
    from opticalflowdiffusion_amd._lib import lib, check, ptr, stream
    g = torch.Generator(device=dev).manual_seed(4321)
    img4 = torch.rand(B, 4, H, W, device=dev, generator=g)
    img3 = img4[:, :3].contiguous()
    flow = smooth_flow(B, H, W, dev, g)
    out4, out3, mask = torch.empty_like(img4), torch.empty_like(img3), torch.empty_like(img3)
    ws = torch.empty(lib().ofd_splat_workspace_bytes(B, H, W), dtype=torch.uint8, device=dev)
    px = float(B * H * W)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    res = {"shape": [B, H, W], "flow": "N(0,8^2) px, 9x9 box filter, clamp +-20 (SURVEY 8d)", "launches_timed": reps}
    ms = timed(lambda: check(lib().ofd_splat_fwd(ptr(img4), ptr(flow), ptr(out4), B, 4, H, W, 1, 0, 0, WARP_RADIUS, ptr(ws), ws.numel(), stream())))
    res["splat_fwd"] = {"entry": "ofd_splat_fwd (softsplat_out SS:352-423, C=3+1, scale 1)", "ms": ms, "algorithmic_bytes": 40.0 * px,
                        "GBps": 40.0 * px / ms / 1e6, "frac_of_hbm_peak": 40.0 * px / ms / 1e6 / PEAK_HBM_GBS}
    ms = timed(lambda: check(lib().ofd_grid_warp_fwd(ptr(img3), ptr(flow), ptr(out3), ptr(mask), B, 3, H, W, stream())))
    res["grid_warp_fwd"] = {"entry": "ofd_grid_warp_fwd (warp_backward_flow WP:95-119: 2x grid_sample + mask, C=3)", "ms": ms,
                            "algorithmic_bytes": 44.0 * px, "GBps": 44.0 * px / ms / 1e6, "frac_of_hbm_peak": 44.0 * px / ms / 1e6 / PEAK_HBM_GBS}
    assert torch.isfinite(out4).all() and torch.isfinite(out3).all()
    return res, (img4[:1].cpu(), img3[:1].cpu(), flow[:1].cpu())


def warp_cpu_baseline(sample, B, H, W):
    """the oracle's splat (oracle/splat_ref.c, scalar C, 1 core) and the reference's own CPU op for the grid_sample warp
    (torch F.grid_sample through oracle/warp_ref.py) on ONE sample of the warp leg's inputs"""
    from oracle import warp_ref as WR
    img4, img3, flow = sample
    WR.splat_out(img4[:, :, :32, :32], flow[:, :, :32, :32])          # builds / loads the C library outside the timed region
    t0 = time.time()
    WR.splat_out(img4, flow)
    dt_s = time.time() - t0
    WR.warp(None, img3, flow, mode="backward")
    reps = 3
    t0 = time.time()
    for _ in range(reps):
        WR.warp(None, img3, flow, mode="backward")
    dt_g = (time.time() - t0) / reps
    px = float(H * W)
    return {"splat_fwd": {"ms_per_sample": dt_s * 1e3, "GBps": 40.0 * px / dt_s / 1e9, "cores": 1, "kind": "port (oracle/splat_ref.c)"},
            "grid_warp_fwd": {"ms_per_sample": dt_g * 1e3, "GBps": 44.0 * px / dt_g / 1e9, "cores": torch.get_num_threads(),
                              "kind": "port (torch CPU grid_sample, the op the reference calls)"},
            "sample": f"1 of {B} samples at {H}x{W}"}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process (python -m torch.distributed.run, one rank
    per GPU) and exit with its code.  Runs before anything touches the GPU in this process (the reference gets its ranks from
    Lightning's devices="auto", experiments/exp_base.py:193-206)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--height", type=int, default=436)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the instrumented loop (per-kernel HIP events) behind the headline loop: no `roofline` object")
    ap.add_argument("--profile-steps", type=int, default=5, help="steps of the instrumented loop")
    ap.add_argument("--no-warp", action="store_true", help="skip the flow-warp leg (`warp` object)")
    ap.add_argument("--dump-launches", default=None, help="CSV path: one row per kernel launch of the instrumented loop")
    ap.add_argument("--train-steps", type=int, default=3, help="timed training steps of the extra `train` object (0: skip)")
    ap.add_argument("--train-warmup", type=int, default=1)
    ap.add_argument("--grad-bucket-dtype", choices=["fp32", "bf16"], default="fp32", help="all-reduce bucket dtype of the training leg (N > 1)")
    ap.add_argument("--with-train", action="store_true", help="(accepted for compatibility: the training leg runs for every N)")
    ap.add_argument("--train-timeout", type=float, default=300.0, help="watchdog of the training leg, seconds")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))           # the parent never initialises the GPU

    from opticalflowdiffusion_amd import parallel
    rank, local_rank, world = parallel.env_rank_world()
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
    torch.manual_seed(parallel.rank_seed(0, rank))
    g = torch.Generator(device="cpu").manual_seed(parallel.rank_seed(1000, rank))
    unet = Unet(64, channels=5, out_dim=2, precision="bf16").to(dev)
    diff = ConditionalDiffusion(unet, (H, W), objective="pred_x0", channels=2, auto_normalize=False,
                                noise_space="image", timesteps=1000, min_snr_loss_weight=True).to(dev)
    cond = (torch.rand(B, 3, H, W, generator=g) * 2 - 1).to(dev)
    img = torch.randn(B, 2, H, W, generator=g).to(dev)

    def step(img, t):
        # noise=None: p_sample draws randn_like(x) itself, inside the step, as the reference does (DD:687)
        out, _, _ = diff.p_sample(img, t, None, external_cond=cond)
        return out

    def sync():
        torch.cuda.synchronize()
        parallel.barrier(dev)

    T = 999
    with torch.no_grad():
        for i in range(args.warmup):
            img = step(img, T - i)
        # ---- headline loop: un-instrumented (no per-kernel events), exactly `steps` steps between two barrier + synchronize pairs
        unet.set_profiling(False)
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            img = step(img, T - args.warmup - i)
        sync()
        elapsed = time.perf_counter() - t0
        assert torch.isfinite(img).all(), "non-finite samples"
        # ---- instrumented loop (feeds `roofline` and `kernel_ms_per_step` only): HIP events around every launch, on the launch stream
        prof, prof_steps, prof_elapsed = {}, max(1, min(args.profile_steps, args.steps)), None
        if not args.no_profile:
            unet.set_profiling(True, args.dump_launches)
            unet.profile(reset=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(prof_steps):
                img = step(img, max(T - args.warmup - args.steps - i, 1))
            torch.cuda.synchronize()
            prof_elapsed = time.perf_counter() - t1
            prof = unet.profile()
            unet.set_profiling(False)

    elapsed = parallel.max_over_ranks(elapsed, dev)

    warp, warp_sample = None, None
    if not args.no_warp and rank == 0:
        try:
            warp, warp_sample = warp_leg(B, H, W, dev)
        except Exception as e:
            warp = {"error": f"{type(e).__name__}: {e}"}

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
            "config": {"workload": workload_name(B, H, W, args.height, args.width),
                       "batch_per_gpu": B, "global_batch": B * world, "height": H, "width": W, "parallelism": f"replicas x{world}"},
            "unet_mfma_frac_of_peak": steps_per_s / world * B * UNET_GFLOP_PER_SAMPLE * (H * W) / (440 * 1024) / 1e3 / PEAK_BF16_TFLOPS,
            "hip_events_in_timed_region": False,
        }
        if warp is not None:
            line["warp"] = warp
        if prof:
            line.update(roofline_objects(prof, prof_steps, prof_elapsed, B, H, W))

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
        del img
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
        if warp_sample is not None:
            try:
                line["cpu_baseline"]["warp"] = warp_cpu_baseline(warp_sample, B, H, W)
            except Exception as e:
                line["cpu_baseline"]["warp"] = {"error": f"{type(e).__name__}: {e}"}
    emit(train, False)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
