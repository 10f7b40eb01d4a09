#!/bin/bash
# same-box A/B of an environment switch: tools/ab_env.sh rounds VAR val1 val2 ...
R=$1; VAR=$2; shift; shift
for r in $(seq $R); do for v in "$@"; do export $VAR=$v
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-steps 0 > gpurun_out/b_${VAR}_$v.json 2>/dev/null; echo -n "$VAR=$v "; python tools/benchsum.py gpurun_out/b_${VAR}_$v.json | head -2 | tr '\n' ' ' | cut -c1-330; echo; done; done
