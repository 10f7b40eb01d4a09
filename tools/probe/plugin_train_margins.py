"""Margins of the training-trajectory assertions of tests/test_plugin_gpu.py (flow: last < 0.7 first over 16 steps; regression: last < 0.8 first over
12; joint: last < first over 10), three repetitions each: the backward's float atomics make trajectories differ run to run."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from opticalflowdiffusion_amd import FlowDiffuser


def run(target, steps, reseed, B, H, W, smooth, **kw):
    torch.manual_seed(0)
    fd = FlowDiffuser(dict(target=target, image_size=[H, W], timesteps=50, flow_max=20, zero_init=False, lr=2e-4, weight_decay=0.0, **kw)).cuda()
    fd.log_dict = lambda *a, **k: None
    opt = fd.configure_optimizers()
    img = torch.rand(B, 3, H, W, device="cuda")
    if smooth:
        flow = torch.clamp(torch.nn.functional.avg_pool2d(torch.randn(B, 2, H, W, device="cuda") * 30, 9, 1, 4), -20, 20)
    else:
        flow = torch.clamp(torch.randn(B, 2, H, W, device="cuda") * 8, -20, 20)
    out = []
    for it in range(steps):
        if reseed:
            torch.manual_seed(100)
        loss = fd.training_step((img, img, flow), it)
        opt.zero_grad(); loss.backward(); opt.step()
        out.append(float(loss.detach()))
    return out


for name, args in (("flow 16 steps (assert last < 0.7 first, max < 1.5 first)", dict(target="flow", steps=16, reseed=True, B=4, H=32, W=64, smooth=False)),
                   ("joint 10 steps (assert last < first)", dict(target="joint", steps=10, reseed=True, B=2, H=32, W=64, smooth=True))):
    for rep in range(3):
        l = run(**args)
        print(json.dumps({"case": name, "rep": rep, "first": round(l[0], 4), "last": round(l[-1], 4), "max": round(max(l), 4), "last_over_first": round(l[-1] / l[0], 3)}), flush=True)
