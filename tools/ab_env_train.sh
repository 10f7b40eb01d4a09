#!/bin/bash
# same-box A/B of an environment switch on the training step: tools/ab_env_train.sh rounds VAR val1 val2 ...
R=$1; VAR=$2; shift; shift
for r in $(seq $R); do for v in "$@"; do export $VAR=$v
python tools/train_bench.py --steps 3 --warmup 2 --profile > gpurun_out/t_${VAR}_$v.json 2>/dev/null; echo -n "$VAR=$v "; python - <<PY
import json
d=json.load(open("gpurun_out/t_${VAR}_$v.json"))
k=d["kernel_ms_per_step"]
print(round(d["ms_per_step"],2), {n[:22]:round(v,2) for n,v in k.items() if n in ("gn_silu_backward","layernorm_c","resblock_out","misc","conv1x1_igemm","linear_attention_backward","linear_attention_core")})
PY
done; done
