set -e
for i in 1 2; do python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests_$i.log 2>&1 || { tail -30 gpurun_out/r03_gpu_tests_$i.log; exit 1; }; tail -1 gpurun_out/r03_gpu_tests_$i.log; done
