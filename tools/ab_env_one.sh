# same-box A/B of one environment switch on the training step: bash tools/ab_env_one.sh rounds VAR "class substrings"
R=$1; V=$2; K=$3
for r in $(seq $R); do for v in 1 0; do
env $V=$v python tools/train_bench.py --steps 3 --warmup 2 --profile > gpurun_out/te_$v.json 2>/dev/null; echo -n "$V=$v "; K="$K" python - <<PY
import json, os
d=json.load(open("gpurun_out/te_$v.json")); k=d["kernel_ms_per_step"]; keys=os.environ["K"].split(",")
print(round(d["ms_per_step"],2), {n[:24]:round(x,2) for n,x in k.items() if any(s in n for s in keys)})
PY
done; done
