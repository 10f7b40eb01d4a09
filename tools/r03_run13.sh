set -e
python -m pytest tests/test_unet_gpu.py -x -q -k "conv1x1 or unet_forward or reference_module" > gpurun_out/r03_t13.log 2>&1 || { tail -40 gpurun_out/r03_t13.log; exit 1; }
tail -1 gpurun_out/r03_t13.log
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-steps 0 --no-warp > gpurun_out/b_$1.json 2>/dev/null; echo -n "$1 "; python - <<PY
import json
d=json.load(open("gpurun_out/b_$1.json")); k=d["kernel_ms_per_step"]
print(round(d["ms_per_step"],2), {n[:16]:round(x,2) for n,x in k.items() if "conv1x1_igemm" in n})
PY
}
for r in 1 2 3; do unset OFD_LIB; run cur; export OFD_LIB=$PWD/opticalflowdiffusion_amd/lib/libofd_hip_c1prev.so; run prev; done
