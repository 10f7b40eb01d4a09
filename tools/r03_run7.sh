set -e
python -m pytest tests/test_unet_gpu.py -x -q -k "unet or linear_attention" > gpurun_out/r03_t7.log 2>&1 || { tail -40 gpurun_out/r03_t7.log; exit 1; }
tail -2 gpurun_out/r03_t7.log
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-steps 0 --no-warp > gpurun_out/b_$1.json 2>/dev/null; echo -n "$1 "; python - <<PY
import json
d=json.load(open("gpurun_out/b_$1.json")); k=d["kernel_ms_per_step"]
print(round(d["ms_per_step"],2), {n[:22]:round(x,2) for n,x in k.items() if "linear_attention_core" in n or "wp_kernel<4" in n})
PY
}
for r in 1 2 3; do
  unset OFD_LIB; unset OFD_LA_DEFER; run cur
  OFD_LA_DEFER=0 run nodefer
  unset OFD_LA_DEFER; export OFD_LIB=$PWD/opticalflowdiffusion_amd/lib/libofd_hip_ln2p.so; run ln2p
done
