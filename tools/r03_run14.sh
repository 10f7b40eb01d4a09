set -e
python -m pytest tests/test_unet_gpu.py tests/test_backward_gpu.py -x -q -k "conv3x3 or unet" > gpurun_out/r03_t14.log 2>&1 || { tail -40 gpurun_out/r03_t14.log; exit 1; }
tail -1 gpurun_out/r03_t14.log
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-steps 0 --no-warp > gpurun_out/b_$1.json 2>/dev/null; echo -n "$1 "; python - <<PY
import json
d=json.load(open("gpurun_out/b_$1.json")); k=d["kernel_ms_per_step"]
print(round(d["ms_per_step"],2), {n[:30]:round(x,2) for n,x in k.items() if "wp_kernel<2" in n})
PY
}
for r in 1 2 3; do OFD_CONV_WP_PERS=4 run pers; OFD_CONV_WP_PERS=0 run nopers; done
