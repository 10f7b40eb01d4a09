"""Which torch-side ops (fills, copies, ...) one training step launches: torch.profiler over one step of tools/train_bench.py's loop."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflowdiffusion_amd import FlowDiffuser
from opticalflowdiffusion_amd import parallel as P
dev = torch.device("cuda", 0); torch.cuda.set_device(dev); P.init(device=dev); torch.manual_seed(0)
B, H, W = 4, 128, 256
fd = FlowDiffuser(dict(target="flow", image_size=[H, W], timesteps=1000, flow_max=20, zero_init=False, lr=1e-4, weight_decay=0.0, clip=100.0)).to(dev)
fd.log_dict = lambda *x, **k: None
opt = fd.configure_optimizers()
img = torch.rand(B, 3, H, W, device=dev); flow = torch.randn(B, 2, H, W, device=dev)
def step(i):
    loss = fd.training_step((img, img, flow), i); opt.zero_grad(); loss.backward(); opt.step(); return loss
for i in range(2): step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step(2)
torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=4).table(sort_by="count", row_limit=25, max_name_column_width=40, max_src_column_width=90))
