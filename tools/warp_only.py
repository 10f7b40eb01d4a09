"""forward splat and grid_sample warp at the BASELINE size, a few launches each (for rocprofv3 passes)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opticalflowdiffusion_amd as m
from opticalflowdiffusion_amd.softsplat import splat_forward
B, H, W = 16, 440, 1024
torch.manual_seed(0)
img4 = torch.rand(B, 4, H, W, device="cuda")
img3 = img4[:, :3].contiguous()
flow = torch.nn.functional.avg_pool2d(torch.randn(B, 2, H, W, device="cuda") * 72, 9, 1, 4).clamp(-20, 20)
for _ in range(4):
    splat_forward(img4, flow)
    m.warp(None, img3, flow, mode="backward")
torch.cuda.synchronize()
