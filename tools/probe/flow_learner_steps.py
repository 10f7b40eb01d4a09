"""Loss sequences of the two FlowLearner training checks of tests/test_flow_learner_gpu.py over more steps / learning rates / repetitions
(the backward has float atomics: trajectories differ run to run) -- to choose assertions that hold with margin."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from opticalflowdiffusion_amd import FlowLearner, warp


def run(kind, lr, steps, seed):
    torch.manual_seed(seed)
    B, H, W = 2, 32, 48
    if kind == "loop":
        fl = FlowLearner(dict(image_size=[H, W], flow_max=20, zero_init=False, lr=lr, weight_decay=0.0, levels=[1, 2, 4], pyramid="loop")).cuda()
        img = torch.rand(B, 3, H, W, device="cuda")
        true_flow = torch.zeros(B, 2, H, W, device="cuda"); true_flow[:, 0] = 3.0
        tgt = torch.nan_to_num(warp(img, None, true_flow, mode="forward"), nan=0.5)
        batch = (img, tgt, true_flow)
    else:
        img = (torch.rand(B, 3, H, W) * 2 - 1).cuda(); tgt = (torch.rand(B, 3, H, W) * 2 - 1).cuda()
        flow = ((torch.rand(B, 2, H, W) * 2 - 1) * 6.0).cuda()
        fl = FlowLearner(dict(image_size=[H, W], flow_max=20, zero_init=False, lr=lr, weight_decay=0.0)).cuda()
        batch = ((img + 1) / 2, (tgt + 1) / 2, flow)
    fl.log_dict = lambda *a, **k: None
    fl.log = lambda *a, **k: None
    opt = fl.configure_optimizers()
    out = []
    for it in range(steps):
        loss = fl.training_step(batch, it)
        opt.zero_grad(); loss.backward(); opt.step()
        out.append(round(float(loss.detach()), 5))
    return out


for kind, seed in (("loop", 0), ("fused", 33)):
    for lr in (5e-5, 2e-4):
        for rep in range(3):
            print(json.dumps({"kind": kind, "lr": lr, "rep": rep, "losses": run(kind, lr, 12, seed)}), flush=True)
