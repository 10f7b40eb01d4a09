"""Margins of the descent-direction check of tests/test_flow_learner_gpu.py: achieved / predicted decrease of one plain gradient step on
FlowLearner's loss, over step sizes, for white-noise and for smooth image pairs, loop and fused pyramids."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from opticalflowdiffusion_amd import FlowLearner, warp
from test_flow_learner_gpu import descent_check

B, H, W = 2, 32, 48
for kind in ("loop", "fused"):
    for smooth in (0, 1):
        torch.manual_seed(0)
        cfg = dict(image_size=[H, W], flow_max=20, zero_init=False, lr=5e-5, weight_decay=0.0)
        if kind == "loop":
            cfg.update(levels=[1, 2, 4], pyramid="loop")
        fl = FlowLearner(cfg).cuda()
        fl.log_dict = lambda *a, **k: None
        fl.log = lambda *a, **k: None
        if smooth:
            img = torch.nn.functional.interpolate(torch.rand(B, 3, H // 8, W // 8, device="cuda"), size=(H, W), mode="bicubic", align_corners=False).clamp(0, 1)
        else:
            img = torch.rand(B, 3, H, W, device="cuda")
        true_flow = torch.zeros(B, 2, H, W, device="cuda"); true_flow[:, 0] = 3.0
        tgt = torch.nan_to_num(warp(img, None, true_flow, mode="forward"), nan=0.5)
        for rep in range(2):
            l0, r = descent_check(fl, (img, tgt, true_flow), fracs=(0.0005, 0.002, 0.005, 0.02, 0.05))
            print(json.dumps({"pyramid": kind, "smooth_images": smooth, "rep": rep, "loss": l0, "achieved_over_predicted": [(f, round(x, 3)) for f, x in r]}), flush=True)
