#!/usr/bin/env python
"""Thin trainer for the FlowDiffuser plugin: what `python main.py experiment=matrix_flow
algorithm=flow_diffuser` does through Lightning (experiments/exp_base.py:177-214), without Lightning / Hydra / W&B.

    python train.py --steps 100 --set algorithm.target=flow algorithm.image_size=[128,256]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 train.py ...

* config = the reference's YAML groups merged the way its defaults lists do (configurations/experiment/base.yaml +
  matrix_flow.yaml, configurations/algorithm/flow_diffuser.yaml); `--config file.yaml` and `--set a.b=v` override;
* one process per GPU; gradients averaged inside every backward by the bucketed RCCL all-reduce (parallel.py);
* checkpoints in Lightning's layout: {"state_dict" (keys unet.* / _model.* / model.*, 13 schedule buffers),
  "optimizer_states", "global_step", "epoch"}; `--resume` continues bit-identically on the same seeds;
* one JSON line per logged step on rank 0.
"""
import argparse
import contextlib
import json
import os
import time

import torch
import yaml

from opticalflowdiffusion_amd import FlowDiffuser, FlowLearner, parallel
from opticalflowdiffusion_amd.data import SintelPairs, SyntheticFlowPairs

DEFAULTS = {
    # configurations/experiment/base.yaml + matrix_flow.yaml
    "experiment": {"name": "matrix_flow", "epochs": -1,
                   "training": {"precision": "bf16", "clipping": 100, "data": {"batch_size": 16, "shuffle": True},
                                "optim": {"accumulate_grad_batches": 1}, "checkpointing": {"every_n_train_steps": 5000}}},
    # configurations/algorithm/flow_diffuser.yaml (+ the non-square image_size extension)
    "algorithm": {"name": "flow_diffuser", "image_size": [128, 256], "latent_dim": 16, "flow_max": 20, "latent_max": 2, "lr": 1e-5,
                  "flow_weight": 0.0, "weight_decay": 1e-6, "is_diffusion": True, "latent": False, "timesteps": 1000,
                  "target": "joint", "ae": "px8q8g0m", "noiser": "image", "zero_init": True},
    "dataset": {"name": "synthetic", "length": 1 << 20, "seed": 0},
}
# configurations/algorithm/flow_learner.yaml; selected with --set algorithm.name=flow_learner (experiments/exp_99.py:22-28)
FLOW_LEARNER = {"name": "flow_learner", "image_size": [128, 128], "flow_max": 20, "latent": False, "zero_init": True, "c2f": False, "lr": 8e-5,
                "weight_decay": 1e-6, "sparsity_weight": 0.0, "occlusion_mask": True, "train_aug": True}
ALGORITHMS = {"flow_diffuser": FlowDiffuser, "flow_learner": FlowLearner}


def deep_update(d, u):
    for k, v in u.items():
        if isinstance(v, dict) and isinstance(d.get(k), dict):
            deep_update(d[k], v)
        else:
            d[k] = v
    return d


def set_path(d, dotted, value):
    keys = dotted.split(".")
    for k in keys[:-1]:
        d = d.setdefault(k, {})
    d[keys[-1]] = yaml.safe_load(value)


def save_checkpoint(path, fd, opt, step, epoch):
    tmp = path + ".tmp"
    torch.save({"state_dict": fd.state_dict(), "optimizer_states": [opt.state_dict()], "global_step": step, "epoch": epoch,
                "pytorch-lightning_version": "compat", "rng": torch.get_rng_state(), "cuda_rng": torch.cuda.get_rng_state()}, tmp)
    os.replace(tmp, path)


def load_checkpoint(path, fd, opt):
    ck = torch.load(path, map_location="cpu", weights_only=False)
    fd.load_state_dict(ck["state_dict"])
    if opt is not None and ck.get("optimizer_states"):
        opt.load_state_dict(ck["optimizer_states"][0])
    if "rng" in ck:
        torch.set_rng_state(ck["rng"])
        torch.cuda.set_rng_state(ck["cuda_rng"].cpu() if torch.is_tensor(ck["cuda_rng"]) else ck["cuda_rng"])
    return int(ck.get("global_step", 0)), int(ck.get("epoch", 0))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=None)
    ap.add_argument("--set", nargs="*", default=[], metavar="a.b=v")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--log-every", type=int, default=10)
    ap.add_argument("--ckpt-dir", default=None)
    ap.add_argument("--resume", default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--config-dir", default=None, help="a `configurations/` tree laid out as the reference's (config.yaml + experiment/ "
                    "algorithm/ dataset/ groups): composed the way main.py's @hydra.main does, then overlaid on the built-in defaults")
    ap.add_argument("--grad-bucket-dtype", choices=["fp32", "bf16"], default="fp32",
                    help="--ddp hook: dtype of the all-reduce buckets (bf16 halves the bytes on the wire; the flat gradient buffer stays fp32)")
    ap.add_argument("--ddp", choices=["hook", "torch"], default="hook", help="multi-GPU gradient averaging: the executor's bucketed "
                    "all-reduce hook (default) or torch.nn.parallel.DistributedDataParallel around the module (what Lightning's DDPStrategy does, exp_base.py:198)")
    ap.add_argument("overrides", nargs="*", help="Hydra-style overrides for --config-dir: group=option, a.b=value, +a.b=value, ~a.b")
    a = ap.parse_args(argv)

    cfg = json.loads(json.dumps(DEFAULTS))
    if a.config_dir:
        from opticalflowdiffusion_amd.compat import compose
        composed = compose(a.config_dir, a.overrides).to_container()
        if composed.get("algorithm", {}).get("name") == "flow_learner":
            cfg["algorithm"] = dict(FLOW_LEARNER)
        deep_update(cfg, composed)
        size = cfg["algorithm"].get("image_size")
        if isinstance(size, str):                          # dataset/sintel.yaml writes "512,256" (W,H)
            w, h = (int(v) for v in size.split(","))
            cfg["algorithm"]["image_size"] = [h, w]
    if a.config:
        deep_update(cfg, yaml.safe_load(open(a.config)) or {})
    if any(kv.replace(" ", "") == "algorithm.name=flow_learner" for kv in a.set):
        cfg["algorithm"] = dict(FLOW_LEARNER)
    for kv in a.set:
        k, v = kv.split("=", 1)
        set_path(cfg, k, v)
    alg, tr = cfg["algorithm"], cfg["experiment"]["training"]
    alg.setdefault("precision", "bf16" if str(tr.get("precision", "bf16")).startswith(("bf16", "16")) else "fp32")
    alg.setdefault("clip", float(tr.get("clipping") or 0.0))          # gradient_clip_val folded into the fused Adam
    H, W = alg["image_size"] if isinstance(alg["image_size"], (list, tuple)) else (alg["image_size"],) * 2

    rank, local_rank, world = parallel.env_rank_world()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    parallel.init(device=dev)
    torch.manual_seed(parallel.rank_seed(a.seed, rank))

    fd = ALGORITHMS[alg.get("name", "flow_diffuser")](alg).to(dev)
    parallel.broadcast_parameters(fd)
    step_module = fd
    if world > 1 and a.ddp == "hook":
        parallel.attach_grad_sync(fd, bucket_dtype=a.grad_bucket_dtype)
    elif world > 1:
        step_module = parallel.TorchDDP(fd, dev)
    opt = fd.configure_optimizers()
    step, epoch = (load_checkpoint(a.resume, fd, opt) if a.resume else (0, 0))

    B = int(tr["data"]["batch_size"])
    ds = SyntheticFlowPairs(cfg["dataset"].get("length", 1 << 20), H, W, flow_max=float(alg["flow_max"]), seed=cfg["dataset"].get("seed", 0))
    if cfg["dataset"].get("name") == "sintel":        # --set dataset.name=sintel dataset.root=/path/to/MPI_Sintel
        files = SintelPairs(cfg["dataset"]["root"], render=cfg["dataset"].get("render", "clean"), image_size=(H, W))

        class _Batches:
            def batch(self, first, count, device):
                items = [files[(first + k) % len(files)] for k in range(count)]
                return tuple(torch.stack(t).to(device, non_blocking=True) for t in zip(*items))
        ds = _Batches()
    accum = int(tr["optim"].get("accumulate_grad_batches", 1))
    every = int((tr.get("checkpointing") or {}).get("every_n_train_steps", 0) or 0)
    t_last, logs = time.perf_counter(), []
    while step < a.steps:
        opt.zero_grad()
        for micro in range(accum):
            base = ((step * accum + micro) * world + rank) * B            # disjoint samples per rank and step
            batch = ds.batch(base, B, dev)
            # torch DDP: gradients are all-reduced by the LAST micro-batch's backward only (no_sync on the others)
            hold = step_module.no_sync() if (micro + 1 < accum and hasattr(step_module, "no_sync")) else contextlib.nullcontext()
            with hold:
                loss = step_module.training_step(batch, step)
                (loss / accum).backward()
        opt.step()
        step += 1
        if rank == 0 and (step % a.log_every == 0 or step == a.steps):
            torch.cuda.synchronize()
            now = time.perf_counter()
            rec = {"step": step, "loss": float(loss.detach()), "grad_norm": float(opt.last_grad_norm) if opt.last_grad_norm is not None else None,
                   "s_per_step": (now - t_last) / min(a.log_every, step), "world": world, "global_batch": B * world * accum}
            t_last = now
            logs.append(rec)
            print(json.dumps(rec), flush=True)
        if a.ckpt_dir and every and step % every == 0 and rank == 0:
            os.makedirs(a.ckpt_dir, exist_ok=True)
            save_checkpoint(os.path.join(a.ckpt_dir, f"step={step}.ckpt"), fd, opt, step, epoch)
        if a.ckpt_dir and every and step % every == 0:
            parallel.barrier(dev)
    if a.ckpt_dir and rank == 0:
        os.makedirs(a.ckpt_dir, exist_ok=True)
        save_checkpoint(os.path.join(a.ckpt_dir, "last.ckpt"), fd, opt, step, epoch)
    parallel.barrier(dev)
    return fd, logs


if __name__ == "__main__":
    main()
