"""hipGraph replay vs per-kernel launches of the denoise step (UNet forward + DDPM update), at the reference's default
image size (flow_diffuser.yaml: 128) where the step is launch-bound, and at the benchmark size where it is not."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflowdiffusion_amd import Unet, ConditionalDiffusion


def run(B, H, W, steps, graph):
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    unet = Unet(64, channels=5, out_dim=2).to(dev)
    diff = ConditionalDiffusion(unet, (H, W), objective="pred_x0", channels=2, auto_normalize=False, timesteps=1000, min_snr_loss_weight=True).to(dev)
    unet.set_graph(graph)
    cond, img = torch.rand(B, 3, H, W, device=dev) * 2 - 1, torch.randn(B, 2, H, W, device=dev)
    noise = torch.randn(B, 2, H, W, device=dev)
    with torch.no_grad():
        for i in range(3):
            img, _, _ = diff.p_sample(img, 999 - i, None, external_cond=cond, noise=noise)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            img, _, _ = diff.p_sample(img, 990 - i, None, external_cond=cond, noise=noise)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for B, H, W, steps in ((16, 128, 128, 100), (1, 128, 128, 100), (16, 440, 1024, 20)):
    e, g = run(B, H, W, steps, False), run(B, H, W, steps, True)
    print(json.dumps({"config": f"denoise step B={B} {H}x{W}", "ms_per_step_launches": round(e, 3), "ms_per_step_graph": round(g, 3), "speedup": round(e / g, 2)}))
