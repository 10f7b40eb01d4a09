import os, sys, subprocess
code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from opticalflowdiffusion_amd.softsplat import splat_forward
B, H, W = 16, 440, 1024
torch.manual_seed(0)
img4 = torch.rand(B, 4, H, W, device="cuda")
flow = torch.nn.functional.avg_pool2d(torch.randn(B, 2, H, W, device="cuda") * 72, 9, 1, 4).clamp(-20, 20)
for _ in range(3): splat_forward(img4, flow)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): splat_forward(img4, flow)
e1.record(); torch.cuda.synchronize()
print("dbg", os.environ.get("OFD_SPLAT_DBG"), "%.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
'''
for d in (0, 1, 2, 3, 4, 7):
    env = dict(os.environ, OFD_SPLAT_DBG=str(d))
    subprocess.run([sys.executable, "-c", code], env=env)
