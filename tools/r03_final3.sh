# closing check of the round: full GPU suite + the default bench line
set -e
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r03_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r03_gpu_tests.log
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python tools/benchsum.py gpurun_out/bench_default.json | head -3 | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
