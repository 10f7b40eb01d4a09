"""FlowLearner's photometric pyramid (flow_learner.py:159-206): the reference's structure (L*L splats per level and image, 1052
splat pairs per step) against the fused pyramid splat (one scale-1 splat + tent filter + border scatter per level and image).
Prints one JSON line per configuration: forward+backward time of the loss alone (no UNet)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


def main():
    from opticalflowdiffusion_amd.flow_learner import LEVELS, photometric_pyramid_loss, photometric_pyramid_loss_fused
    cases = [(16, 128, 128, 1), (16, 440, 1024, 0)]
    if "--full-only" in sys.argv:
        cases = cases[1:]
    for (B, H, W, n_loop) in cases:
        torch.manual_seed(0)
        img = torch.rand(B, 3, H, W, device="cuda") * 2 - 1
        tgt = torch.rand(B, 3, H, W, device="cuda") * 2 - 1
        flow = torch.nn.functional.avg_pool2d(torch.randn(B, 2, H, W, device="cuda") * 72, 9, 1, 4).clamp(-20, 20).requires_grad_(True)
        wts = (torch.randn(B, 1, H, W, device="cuda") * 0.3).requires_grad_(True)

        def step(fn):
            flow.grad = None
            wts.grad = None
            fn(img, flow, wts, tgt, LEVELS).backward()
        row = {"workload": f"photometric pyramid loss fwd+bwd, B={B} {H}x{W}, levels {list(LEVELS)} (1052 offsets)",
               "fused_ms": timed(lambda: step(photometric_pyramid_loss_fused), 3)}
        if n_loop:
            row["loop_ms"] = timed(lambda: step(photometric_pyramid_loss), n_loop)
            row["speedup"] = row["loop_ms"] / row["fused_ms"]
        print(json.dumps(row), flush=True)


def learner_step():
    """whole FlowLearner training step (regression UNet forward + backward, pyramid loss, fused Adam) at the reference's default size"""
    from opticalflowdiffusion_amd import FlowLearner
    from opticalflowdiffusion_amd.data import SyntheticFlowPairs
    for (B, H, W) in [(16, 128, 128)]:
        fl = FlowLearner(dict(image_size=[H, W], flow_max=20, zero_init=False)).cuda()
        fl.log_dict = lambda *a, **k: None
        fl.log = lambda *a, **k: None
        opt = fl.configure_optimizers()
        batch = SyntheticFlowPairs(64, H, W).batch(0, B, torch.device("cuda"))

        def step():
            opt.zero_grad()
            fl.training_step(batch, 0).backward()
            opt.step()
        print(json.dumps({"workload": f"FlowLearner training step, B={B} {H}x{W}, 10 levels, fused pyramid", "ms_per_step": timed(step, 5)}), flush=True)


if __name__ == "__main__":
    main()
    if "--full-only" not in sys.argv:
        learner_step()
