set -e
python -m pytest tests/test_unet_gpu.py tests/test_backward_gpu.py -x -q -k "unet or gn_ or resblock or building" > gpurun_out/r03_t8.log 2>&1 || { tail -40 gpurun_out/r03_t8.log; exit 1; }
tail -2 gpurun_out/r03_t8.log
for r in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-steps 0 --no-warp > gpurun_out/b_gnf.json 2>/dev/null; python - <<PY
import json
d=json.load(open("gpurun_out/b_gnf.json")); k=d["kernel_ms_per_step"]
print(round(d["ms_per_step"],2), {n[:22]:round(x,3) for n,x in k.items() if x>0})
PY
done
