"""Per-sample time of the denoise step against the batch size (does a batch whose tensors fit the 256 MB Infinity Cache run faster per sample?)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from opticalflowdiffusion_amd import Unet, ConditionalDiffusion
    H, W = 440, 1024
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    unet = Unet(64, channels=5, out_dim=2, precision="bf16").to(dev)
    diff = ConditionalDiffusion(unet, (H, W), objective="pred_x0", channels=2, auto_normalize=False, noise_space="image",
                                timesteps=1000, min_snr_loss_weight=True).to(dev)
    unet.set_split_streams(False)
    for B in (16, 8, 4, 2, 1, 16):
        cond = torch.rand(B, 3, H, W, device=dev) * 2 - 1
        x0 = torch.randn(B, 2, H, W, device=dev)
        noise = torch.randn(B, 2, H, W, device=dev)
        reps = 16 // B
        with torch.no_grad():
            for _ in range(2 * reps):
                diff.p_sample(x0, 999, None, external_cond=cond, noise=noise)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 4
            for _ in range(n * reps):
                diff.p_sample(x0, 999, None, external_cond=cond, noise=noise)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
        print(json.dumps({"B": B, "ms_per_16_samples": round(ms, 3)}), flush=True)


if __name__ == "__main__":
    main()
