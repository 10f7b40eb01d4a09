"""Launch-gap probe: runs the denoise step at 128x128 under rocprofv3 --kernel-trace and reports, from the trace, how the step
time splits into kernel execution and the idle time between consecutive kernels."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflowdiffusion_amd import Unet, ConditionalDiffusion

B, H, W = 16, 128, 128
dev = torch.device("cuda", 0)
unet = Unet(64, channels=5, out_dim=2).to(dev)
diff = ConditionalDiffusion(unet, (H, W), objective="pred_x0", channels=2, auto_normalize=False, timesteps=1000, min_snr_loss_weight=True).to(dev)
cond, img = torch.rand(B, 3, H, W, device=dev) * 2 - 1, torch.randn(B, 2, H, W, device=dev)
noise = torch.randn(B, 2, H, W, device=dev)
with torch.no_grad():
    for i in range(30):
        img, _, _ = diff.p_sample(img, 999 - i, None, external_cond=cond, noise=noise)
    torch.cuda.synchronize()
