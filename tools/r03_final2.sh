set -e
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
bash tools/pmc.sh > gpurun_out/pmc.txt 2>&1 || true
tail -8 gpurun_out/pmc.txt
python bench.py > gpurun_out/bench_default2.json 2> gpurun_out/bench_default2.err
OFD_SPLIT_STREAMS=0 bash tools/prof_infer.sh > gpurun_out/prof_infer_onestream.txt 2>&1 || true
cp gpurun_out/prof_infer_kernel_stats.csv gpurun_out/prof_infer_onestream_kernel_stats.csv; cp gpurun_out/prof_infer_bench.json gpurun_out/prof_infer_onestream_bench.json
bash tools/prof_infer.sh > gpurun_out/prof_infer.txt 2>&1 || true
tail -2 gpurun_out/prof_infer.txt
