#!/bin/bash
# same-box A/B of library builds: tools/ab_libs.sh rounds tag1 tag2 ...  ("cur" = the default library)
R=$1; shift
for r in $(seq $R); do for v in "$@"; do if [ $v = cur ]; then unset OFD_LIB; else export OFD_LIB=$PWD/opticalflowdiffusion_amd/lib/libofd_hip_$v.so; fi
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-steps 0 > gpurun_out/b_$v.json 2>/dev/null; echo -n "$v "; python tools/benchsum.py gpurun_out/b_$v.json | head -2 | tr '\n' ' ' | cut -c1-230; echo; done; done
