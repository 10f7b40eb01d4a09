"""BASELINE configs[2] (C3): 50-step DDIM sampling + flow-warp reconstruction, bs=64, one MI355X.
FlowDiffuser.sample() end to end: 50 UNet forwards + fused DDIM updates + forward splat of the condition image."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflowdiffusion_amd import FlowDiffuser


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--height", type=int, default=440)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    fd = FlowDiffuser(dict(target="flow", image_size=[a.height, a.width], timesteps=1000, sampling_timesteps=a.steps, flow_max=20,
                           zero_init=False)).to(dev)
    cond = torch.rand(a.batch, 3, a.height, a.width, device=dev) * 2 - 1
    flow = torch.zeros(a.batch, 2, a.height, a.width, device=dev)
    with torch.no_grad():
        # warm-up at the full batch: prepares the weights and allocates the workspace (a 96 GB hipMalloc takes seconds)
        fd.unet(torch.randn(a.batch, 2, a.height, a.width, device=dev), external_cond=cond, time=torch.zeros(a.batch, dtype=torch.long, device=dev))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        samples, traj = fd.sample(cond, flow)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert samples.shape == (a.batch, 3, a.height, a.width) and torch.isfinite(traj).all()
    holes = float(torch.isnan(samples).float().mean())
    print(json.dumps({"config": f"C3: {a.steps}-step DDIM + splat reconstruction, bs={a.batch}, {a.height}x{a.width}", "seconds": dt,
                      "samples_per_s": a.batch / dt, "denoise_steps_per_s_at_bs": a.steps / dt, "ms_per_denoise_step": 1e3 * dt / a.steps,
                      "nan_hole_fraction_of_reconstruction": holes, "max_mem_GiB": torch.cuda.max_memory_allocated() / 2 ** 30}))


if __name__ == "__main__":
    main()
