"""`target: joint` training at a small size with the per-level terms of the pyramid loss (DD:893-973) logged:
loss = sum_L L^4 S_L / sum_L N_L.  One JSON line per logged step."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflowdiffusion_amd import FlowDiffuser
from opticalflowdiffusion_amd.data import SyntheticFlowPairs

def main():
    H, W, B = int(os.environ.get("SOAK_H", 64)), int(os.environ.get("SOAK_W", 128)), int(os.environ.get("SOAK_B", 8))
    steps, lr = int(os.environ.get("SOAK_STEPS", 300)), float(os.environ.get("SOAK_LR", 1e-5))
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    fd = FlowDiffuser(dict(target="joint", image_size=[H, W], timesteps=1000, flow_max=20, zero_init=True, lr=lr, weight_decay=1e-6, clip=100.0)).to(dev)
    fd.log_dict = lambda *a, **k: None
    opt = fd.configure_optimizers()
    ds = SyntheticFlowPairs(1 << 20, H, W, flow_max=20.0, seed=0)
    for step in range(steps):
        batch = ds.batch(step * B, B, dev)
        opt.zero_grad()
        loss = fd.training_step(batch, step)
        loss.backward()
        opt.step()
        if step % max(1, steps // 30) == 0 or step == steps - 1:
            lv = {int(L): (float(s) * L ** 4, float(n)) for L, s, n in fd.model.last_levels}
            tot_n = sum(n for _, n in lv.values())
            print(json.dumps({"step": step, "loss": round(float(loss.detach()), 4), "grad_norm": round(float(opt.last_grad_norm), 2),
                              "term_L": {L: round(v / tot_n, 4) for L, (v, n) in lv.items()}, "N_L": {L: int(n) for L, (v, n) in lv.items()},
                              "final_conv_w_norm": round(float(fd.unet.final_conv.weight.norm()), 5)}), flush=True)

if __name__ == "__main__":
    main()
