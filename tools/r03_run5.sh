set -e
python -m pytest tests/test_warp_gpu.py tests/test_flow_learner_gpu.py tests/test_plugin_gpu.py -x -q > gpurun_out/r03_t5.log 2>&1 || { tail -40 gpurun_out/r03_t5.log; exit 1; }
tail -2 gpurun_out/r03_t5.log
for r in 1 2 3; do for f in 1; do
OFD_SPLAT_FAST=$f python - <<PY
import sys,os
sys.path.insert(0,'.')
import torch, bench
w,_=bench.warp_leg(16,440,1024,torch.device('cuda',0),reps=50)
import opticalflowdiffusion_amd as m
img3=torch.rand(16,3,440,1024,device='cuda'); g=torch.Generator(device='cuda').manual_seed(1); flow=bench.smooth_flow(16,440,1024,torch.device('cuda',0),g)
for _ in range(3): m.warp(img3,None,flow,mode='forward')
torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(30): m.warp(img3,None,flow,mode='forward')
e1.record(); torch.cuda.synchronize()
print('fast=%s'%os.environ['OFD_SPLAT_FAST'], 'splat %.2f us %.3f   warp(forward) wrapper %.2f us'%(w['splat_fwd']['ms']*1e3,w['splat_fwd']['frac_of_hbm_peak'], e0.elapsed_time(e1)/30*1e3))
PY
done; done
