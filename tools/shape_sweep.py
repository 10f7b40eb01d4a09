"""Ad-hoc robustness sweep: inference and training forward of the HIP UNet against the bf16c oracle at awkward sizes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import rel_l2  # noqa: E402
from oracle import unet_ref as R  # noqa: E402
from opticalflowdiffusion_amd import Unet  # noqa: E402

torch.manual_seed(0)
net = Unet(64, channels=5, out_dim=2).cuda()
P = {n: p.detach().cpu().clone() for n, p in net.named_parameters()}
bad = 0
for (B, H, W) in [(1, 8, 8), (1, 16, 8), (2, 8, 40), (1, 72, 136), (5, 16, 16), (1, 8, 264), (3, 40, 8), (1, 104, 200)]:
    x = torch.randn(B, 2, H, W)
    cond = torch.rand(B, 3, H, W) * 2 - 1
    t = torch.randint(0, 1000, (B,))
    with torch.no_grad():
        ref = R.unet_forward(P, x, cond, t, mode="bf16c")
        out = net(x.cuda(), cond.cuda(), t.cuda()).cpu()
    out_t = net(x.cuda(), external_cond=cond.cuda(), time=t.cuda())
    out_t.sum().backward()
    gfin = all(torch.isfinite(p.grad).all() for p in net.parameters())
    net.zero_grad()
    e1, e2 = rel_l2(out, ref), rel_l2(out_t.detach().cpu(), ref)
    flag = "" if (e1 < 3e-2 and e2 < 3e-2 and gfin) else "   <-- CHECK"
    bad += bool(flag)
    print(f"B={B} {H}x{W}: inference rel-L2 {e1:.2e}, training-forward rel-L2 {e2:.2e}, grads finite {gfin}{flag}", flush=True)
print("bad cases:", bad)
