# Refreshes a round's measured records under gpurun_out/<tag>_* (copy what should be judged into profiles/):
#     bash tools/refresh_profiles.sh r04
# Every rocprofv3 run puts the program itself after `--` and collects counters in passes of their own (--kernel-trace only).
TAG=${1:-rXX}; PART=${2:-all}      # part: a (denoise leg: bench, rocprof, counters), b (warp, training, C3, C5), all
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
if [ $PART = a ] || [ $PART = all ]; then
# ---- the driver's command, and the same command on one stream (whole-batch launches: what bench.py's instrumented loop times)
python3 bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
echo "bench default: $(python3 tools/benchsum.py $O/${TAG}_bench_default.json | head -1)"
# ---- rocprofv3 --kernel-trace --stats of the denoise leg on ONE stream: the per-kernel averages `roofline` must agree with
rm -rf $O/prof_os
OFD_SPLIT_STREAMS=0 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_os -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-steps 0 --no-warp > $O/prof_os.log 2>&1
grep '"metric"' $O/prof_os.log > $O/${TAG}_bench_denoise_one_stream_under_rocprof.json
cp $(ls $O/prof_os/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats_denoise_one_stream.csv
# ---- the default command (two streams) under rocprof
rm -rf $O/prof_ds
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ds -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-steps 0 > $O/prof_ds.log 2>&1
grep '"metric"' $O/prof_ds.log > $O/${TAG}_bench_denoise_under_rocprof.json
cp $(ls $O/prof_ds/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats_denoise.csv
# ---- PMC passes (HBM bytes, MFMA-busy) and SQ passes of the one-stream denoise leg
bash tools/pmc.sh > $O/pmc.txt 2>&1
for t in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do cp $O/pmc_$t.csv $O/${TAG}_pmc_$t.csv 2>/dev/null; done
cp $O/pmc_summary.json $O/${TAG}_pmc_summary.json 2>/dev/null
OFD_SPLIT_STREAMS=0 bash tools/pmc_sq.sh > $O/pmc_sq.txt 2>&1
cp $O/sq_summary.json $O/${TAG}_sq_counters.json 2>/dev/null
fi
if [ $PART = b ] || [ $PART = all ]; then
# ---- flow-warp kernels: bench + counters
python3 tools/warp_bench.py > $O/${TAG}_warp_bench.jsonl 2>/dev/null
bash tools/pmc_warp.sh > $O/pmc_warp.txt 2>&1
cp $O/pmc_warp_summary.json $O/${TAG}_pmc_warp.json 2>/dev/null
# ---- training step, C3, C5
python3 tools/train_bench.py --steps 5 --warmup 2 --profile > $O/${TAG}_train_bench_profile.json 2>/dev/null
python3 tools/train_bench.py --target joint --height 448 --steps 3 --warmup 2 > $O/${TAG}_train_bench_joint_448.json 2>/dev/null
python3 tools/ddim_bench.py > $O/${TAG}_c3_ddim50_bs64.json 2>/dev/null
rm -rf $O/prof_c5
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python3 bench.py --batch 1 --height 1080 --width 1920 --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_c5.log 2>&1
grep '"metric"' $O/prof_c5.log > $O/${TAG}_c5_1080p_b1.json
cp $(ls $O/prof_c5/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats_c5_1080p_b1.csv
bash tools/prof_train.sh > $O/prof_train.txt 2>&1
cp $O/prof_train_kernel_stats.csv $O/${TAG}_kernel_stats_train.csv 2>/dev/null
fi
echo "done: $(ls $O/${TAG}_* | wc -l) files"
