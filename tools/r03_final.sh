# round-3 records: full GPU suite, the default bench line, rocprof stats of the same command (denoise only + full), PMC passes, C3 / C5 / joint / training profile
set -e
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r03_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r03_gpu_tests.log
bash tools/refresh_profiles.sh
bash tools/prof_infer.sh > gpurun_out/prof_infer.txt 2>&1 || true
tail -3 gpurun_out/prof_infer.txt
bash tools/pmc.sh > gpurun_out/pmc.txt 2>&1 || true
tail -16 gpurun_out/pmc.txt
python tools/warp_bench.py > gpurun_out/warp_bench.jsonl 2>/dev/null || true
