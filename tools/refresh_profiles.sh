# refreshes the round's measured records under gpurun_out/ (copy what should be judged into profiles/): bash tools/refresh_profiles.sh
set -e
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench default done"
python tools/train_bench.py --steps 5 --warmup 2 --profile > gpurun_out/train_bench_profile.json 2>/dev/null
python tools/train_bench.py --target joint --height 448 --steps 3 --warmup 2 > gpurun_out/train_bench_joint_448.json 2>/dev/null
python tools/ddim_bench.py > gpurun_out/c3_ddim50_bs64.json 2>/dev/null
python bench.py --batch 1 --height 1080 --width 1920 --no-cpu-baseline > gpurun_out/c5_1080p_b1.json 2>/dev/null
echo "benches done"
bash tools/prof_final.sh > gpurun_out/prof_final.txt 2>&1
bash tools/prof_train.sh > gpurun_out/prof_train.txt 2>&1
tail -3 gpurun_out/prof_final.txt
