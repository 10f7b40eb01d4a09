set -e
python -m pytest tests/test_unet_gpu.py -x -q -k "upsample or unet_forward or reference_module" > gpurun_out/r03_t6.log 2>&1 || { tail -40 gpurun_out/r03_t6.log; exit 1; }
tail -2 gpurun_out/r03_t6.log
for r in 1 2 3; do for v in 1 0; do OFD_PHASE_WP=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-steps 0 --no-warp > gpurun_out/b_ph$v.json 2>/dev/null; echo -n "phase_wp=$v "; python tools/benchsum.py gpurun_out/b_ph$v.json | head -2 | tr '\n' ' ' | sed 's/conv3x3_wp_kernel/wp/g; s/gn_finalize.*flash_attention_backward=0.00//' | cut -c1-260; echo; done; done
