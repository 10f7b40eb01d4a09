"""Diagnostic: run-to-run reproducibility of FlowDiffuser training, default (float atomics) vs deterministic mode (csrc/det.h).

    python tools/probe/determinism.py [H W B steps [learner-loop | learner-fused]]

Two runs from the same seed per mode; prints how many parameters and losses differ bit-wise after `steps` steps, and the step time."""
import gc, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from opticalflowdiffusion_amd import FlowDiffuser, FlowLearner

H, W, B, steps = (list(map(int, sys.argv[1:5])) + [40, 72, 3, 6][len(sys.argv[1:5]):])[:4]
WHAT = sys.argv[5] if len(sys.argv) > 5 else "diffuser"


def run(det):
    torch.manual_seed(0)
    if WHAT == "diffuser":
        fd = FlowDiffuser(dict(target="flow", image_size=[H, W], timesteps=50, flow_max=20, zero_init=False, lr=2e-4, weight_decay=1e-4, gradient_clip_val=0.5)).cuda()
        unet = fd.unet
    else:
        fd = FlowLearner(dict(image_size=[H, W], flow_max=20, zero_init=False, lr=5e-5, weight_decay=0.0, levels=[1, 2, 4], pyramid=WHAT.split("-")[1])).cuda()
        fd.log = lambda *a, **k: None
        unet = fd.unet.model
    fd.log_dict = lambda *a, **k: None
    unet.set_deterministic(det)
    opt = fd.configure_optimizers()
    g = torch.Generator(device="cuda").manual_seed(5)
    img = torch.rand(B, 3, H, W, device="cuda", generator=g)
    tgt = torch.rand(B, 3, H, W, device="cuda", generator=g)
    flow = torch.clamp(torch.randn(B, 2, H, W, device="cuda", generator=g) * 8, -20, 20)
    losses, t0 = [], None
    for it in range(steps):
        if it == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = fd.training_step((img, tgt, flow), it)
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(loss.detach().clone())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / max(1, steps - 2) * 1e3 if t0 else None
    out = (torch.stack(losses), torch.cat([p.detach().flatten() for p in unet.parameters()]).clone(), ms, unet.deterministic_misses())
    del fd, opt, loss
    gc.collect(); torch.cuda.empty_cache()          # (the 108 GiB training workspace of the full-size run)
    return out


for det in (False, True):
    l1, p1, ms1, _ = run(det)
    l2, p2, ms2, miss = run(det)
    print(json.dumps({"model": WHAT, "deterministic": det, "shape": [B, H, W], "steps": steps, "losses_differing": int((l1 != l2).sum()), "params_differing": int((p1 != p2).sum()),
                      "params": p1.numel(), "max_abs_param_diff": float((p1 - p2).abs().max()), "ms_per_step": [ms1, ms2], "shadow_misses": miss}), flush=True)
