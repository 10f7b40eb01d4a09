# per-launch view of the weight gradients of one training step: bash tools/dump_wgrad.sh [pattern]
rm -f gpurun_out/lw.csv
python tools/train_bench.py --steps 2 --warmup 2 --profile --dump gpurun_out/lw.csv > gpurun_out/tw.json 2>/dev/null
python tools/top_launches.py gpurun_out/lw.csv 400 | grep "${1:-wgrad}" | head -40
