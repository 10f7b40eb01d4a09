set -e
for wp in 3 7; do OFD_CONV_WP=$wp python -m pytest tests/test_flow_learner_gpu.py -x -q -s -k training_reduces 2>&1 | grep -E "FlowLearner losses|passed|failed"; done
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r03_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r03_gpu_tests.log
python bench.py > gpurun_out/r03_bench1.json 2> gpurun_out/r03_bench1.err || { tail -20 gpurun_out/r03_bench1.err; exit 1; }
cat gpurun_out/r03_bench1.json
