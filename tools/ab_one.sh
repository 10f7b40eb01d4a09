# one training-bench line per library build with the classes named on the command line: bash tools/ab_one.sh rounds "class substrings" tag1 tag2 ...
R=$1; K=$2; shift; shift
for r in $(seq $R); do for v in "$@"; do if [ $v = cur ]; then unset OFD_LIB; else export OFD_LIB=$PWD/opticalflowdiffusion_amd/lib/libofd_hip_$v.so; fi
python tools/train_bench.py --steps 3 --warmup 2 --profile > gpurun_out/tl_$v.json 2>/dev/null; echo -n "$v "; K="$K" python - <<PY
import json, os
d=json.load(open("gpurun_out/tl_$v.json")); k=d["kernel_ms_per_step"]; keys=os.environ["K"].split(",")
print(round(d["ms_per_step"],2), {n[:24]:round(x,2) for n,x in k.items() if any(s in n for s in keys)})
PY
done; done
