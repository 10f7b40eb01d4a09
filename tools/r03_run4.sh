for dbg in 0 1 2 4 8 3 6 7 15 14; do
OFD_GW_DBG=$dbg python - <<PY
import sys,os
sys.path.insert(0,'.')
import torch, bench
from opticalflowdiffusion_amd._lib import lib, check, ptr, stream
B,H,W=16,440,1024
dev=torch.device('cuda',0)
g=torch.Generator(device=dev).manual_seed(4321)
img3=torch.rand(B,3,H,W,device=dev,generator=g); flow=bench.smooth_flow(B,H,W,dev,g)
out3,mask=torch.empty_like(img3),torch.empty_like(img3)
def fn(): check(lib().ofd_grid_warp_fwd(ptr(img3),ptr(flow),ptr(out3),ptr(mask),B,3,H,W,stream()))
for _ in range(5): fn()
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): fn()
e1.record(); torch.cuda.synchronize()
print('dbg=%2s  %.2f us'%(os.environ['OFD_GW_DBG'], e0.elapsed_time(e1)/50*1e3))
PY
done
