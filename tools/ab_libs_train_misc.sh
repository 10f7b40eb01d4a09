# tools/ab_libs_train.sh with the small-kernel classes printed: bash tools/ab_libs_train_misc.sh rounds tag1 tag2 ...
R=$1; shift
for r in $(seq $R); do for v in "$@"; do if [ $v = cur ]; then unset OFD_LIB; else export OFD_LIB=$PWD/opticalflowdiffusion_amd/lib/libofd_hip_$v.so; fi
python tools/train_bench.py --steps 3 --warmup 2 --profile > gpurun_out/tl_$v.json 2>/dev/null; echo -n "$v "; python - <<PY
import json
d=json.load(open("gpurun_out/tl_$v.json")); k=d["kernel_ms_per_step"]
print(round(d["ms_per_step"],2), {n[:22]:round(x,2) for n,x in k.items() if n in ("misc","gn_silu_backward","layernorm_c","conv_wgrad_kernel<3>")})
PY
done; done
