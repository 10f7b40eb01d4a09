# A/B of the wide 1x1 weight-gradient kernel on the training step (same box): bash tools/ab_wide.sh
for r in 1 2; do for v in 1 0; do
OFD_NO_WGRAD1_WIDE=$v python tools/train_bench.py --steps 3 --warmup 2 --profile > gpurun_out/tw_$v.json 2>/dev/null; echo -n "no_wide=$v "; python - <<PY
import json
d=json.load(open("gpurun_out/tw_$v.json")); k=d["kernel_ms_per_step"]
print(round(d["ms_per_step"],2), {n[:22]:round(x,2) for n,x in k.items() if "wgrad" in n})
PY
done; done
