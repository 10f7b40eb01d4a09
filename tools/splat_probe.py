import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.warp_bench import timed
from opticalflowdiffusion_amd.softsplat import splat_forward
B, H, W = 16, 440, 1024
torch.manual_seed(0)
img4 = torch.rand(B, 4, H, W, device="cuda")
img1 = img4[:, :1].contiguous()
z = torch.zeros(B, 2, H, W, device="cuda")
sm = (torch.rand(B, 2, H, W, device="cuda") * 2 - 1) * 20
for name, flow in (("uniform20", sm),):
    for r in (24,):
        ms = timed(lambda: splat_forward(img4, flow, radius=r))
        ms1 = timed(lambda: splat_forward(img1, flow, radius=r))
        print(f"flow={name} radius={r}: C=4 {ms:.3f} ms ({40*B*H*W/ms/1e6:.0f} GB/s)   C=1 {ms1:.3f} ms")
