set -e
python -m pytest tests/test_unet_gpu.py -x -q -k "conv3x3 or conv or unet_forward" > gpurun_out/r03_t1.log 2>&1 || { tail -30 gpurun_out/r03_t1.log; exit 1; }
tail -2 gpurun_out/r03_t1.log
bash tools/ab_libs.sh 2 cur nopeel dot2
echo "--- OFD_CONV_WP=7 (wp<2,2> for 64->64 too)"
for r in 1 2; do OFD_CONV_WP=7 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-steps 0 > gpurun_out/b_wp7.json 2>/dev/null; python tools/benchsum.py gpurun_out/b_wp7.json | head -2 | tr '\n' ' ' | cut -c1-230; echo; done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-steps 0 --dump-launches gpurun_out/r03_launches0.csv > /dev/null 2>&1
