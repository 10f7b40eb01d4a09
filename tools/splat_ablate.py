import os, subprocess, sys
code = '''
import sys, os, json, torch
sys.path.insert(0, os.getcwd())
from opticalflowdiffusion_amd.softsplat import splat_forward
B,H,W=16,440,1024
torch.manual_seed(0)
img4=torch.rand(B,4,H,W,device="cuda")
smooth=torch.nn.functional.avg_pool2d(torch.randn(B,2,H,W,device="cuda")*72,9,1,4).clamp(-20,20)
for _ in range(5): splat_forward(img4,smooth)
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): splat_forward(img4,smooth)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("OFD_SPLAT_DBG","0"), os.environ.get("OFD_SPLAT_NO_FAST","0"), round(e0.elapsed_time(e1)/30,4))
'''
for nf in ("0", "1"):
    for d in ("0", "1", "2", "3", "4"):
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OFD_SPLAT_DBG=d, OFD_SPLAT_NO_FAST=nf))
