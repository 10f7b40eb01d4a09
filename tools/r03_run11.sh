set -e
python -m pytest tests/test_unet_gpu.py -x -q -k "unet_forward or reference_module or linear_attention or split" > gpurun_out/r03_t11.log 2>&1 || { tail -40 gpurun_out/r03_t11.log; exit 1; }
tail -1 gpurun_out/r03_t11.log
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-steps 0 --no-warp > gpurun_out/b_$1.json 2>/dev/null; echo -n "$1 "; python - <<PY
import json
d=json.load(open("gpurun_out/b_$1.json")); k=d["kernel_ms_per_step"]
print(round(d["ms_per_step"],2), {n[:26]:round(x,2) for n,x in k.items() if "linear_att" in n})
PY
}
for r in 1 2 3; do unset OFD_LIB; run cur; export OFD_LIB=$PWD/opticalflowdiffusion_amd/lib/libofd_hip_laprev.so; run prev; done
