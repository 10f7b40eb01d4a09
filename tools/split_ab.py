"""Same-box A/B of the two-stream half-batch forward (ofd_unet_set_split_streams) against the one-stream forward at the
BASELINE size: interleaved rounds in ONE process, median / min per arm, and torch.equal of the outputs per sample."""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from opticalflowdiffusion_amd import Unet, ConditionalDiffusion
    B = int(os.environ.get("AB_B", 16))
    H, W = int(os.environ.get("AB_H", 440)), int(os.environ.get("AB_W", 1024))
    steps, rounds = int(os.environ.get("AB_STEPS", 6)), int(os.environ.get("AB_ROUNDS", 4))
    offsets = [int(v) for v in os.environ.get("AB_OFFSETS", "0,1,2,3,5").split(",")]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    unet = Unet(64, channels=5, out_dim=2, precision="bf16").to(dev)
    diff = ConditionalDiffusion(unet, (H, W), objective="pred_x0", channels=2, auto_normalize=False, noise_space="image",
                                timesteps=1000, min_snr_loss_weight=True).to(dev)
    cond = torch.rand(B, 3, H, W, device=dev) * 2 - 1
    x0 = torch.randn(B, 2, H, W, device=dev)
    noise = torch.randn(B, 2, H, W, device=dev)

    def run(n):
        img = x0
        for i in range(n):
            img, _, _ = diff.p_sample(img, 999 - i, None, external_cond=cond, noise=noise)
        return img

    arms = [("one-stream", None)] + [(f"split offset {o}", o) for o in offsets]
    times = {a: [] for a, _ in arms}
    outs = {}
    with torch.no_grad():
        for name, off in arms:                      # warm-up + outputs for the equality check
            unet.set_split_streams(off is not None, off if off is not None else -1)
            outs[name] = run(2).clone()
        for r in range(rounds):
            for name, off in arms:
                unet.set_split_streams(off is not None, off if off is not None else -1)
                run(1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(steps)
                torch.cuda.synchronize()
                times[name].append((time.perf_counter() - t0) / steps * 1e3)
    ref = outs["one-stream"]
    for name, _ in arms:
        print(json.dumps({"arm": name, "ms_median": statistics.median(times[name]), "ms_min": min(times[name]), "all": [round(v, 3) for v in times[name]],
                          "equal_to_one_stream": bool(torch.equal(outs[name], ref)), "max_abs_diff": float((outs[name] - ref).abs().max()),
                          "B": B, "H": H, "W": W}), flush=True)


if __name__ == "__main__":
    main()
