set -e
for ring in 2 1; do OFD_GW_RING=$ring python -m pytest tests/test_warp_gpu.py -x -q -k "grid_sample or c5_warp" > gpurun_out/r03_t3_$ring.log 2>&1 || { tail -40 gpurun_out/r03_t3_$ring.log; exit 1; }; tail -1 gpurun_out/r03_t3_$ring.log; done
for r in 1 2 3; do for ring in 2 1 0; do
OFD_GW_RING=$ring python - <<PY
import json,subprocess,sys,os
sys.path.insert(0,'.')
import torch, bench
w,_=bench.warp_leg(16,440,1024,torch.device('cuda',0),reps=50)
print('ring=%s'%os.environ['OFD_GW_RING'], 'grid_warp %.2f us %.3f'%(w['grid_warp_fwd']['ms']*1e3,w['grid_warp_fwd']['frac_of_hbm_peak']))
PY
done; done
