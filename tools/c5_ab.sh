set -e
python -m pytest tests/test_unet_gpu.py tests/test_plugin_gpu.py tests/test_backward_gpu.py -x -q -m gpu > gpurun_out/t_la.log 2>&1
tail -2 gpurun_out/t_la.log
for t in 128 192 256 384; do
  OFD_LA_GX1_TOTAL=$t python bench.py --batch 1 --height 1080 --width 1920 --no-cpu-baseline --train-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('GX1_TOTAL=$t', round(d['ms_per_step'],2), 'la', round(d['kernel_ms_per_step']['linear_attention_core'],2))"
done
for g in; do
  OFD_LA_GX2=$g python bench.py --batch 1 --height 1080 --width 1920 --no-cpu-baseline --train-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('GX2=$g', round(d['ms_per_step'],2), 'la', round(d['kernel_ms_per_step']['linear_attention_core'],2))"
done
python bench.py --no-cpu-baseline --train-steps 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B16', round(d['ms_per_step'],2), 'la', round(d['kernel_ms_per_step']['linear_attention_core'],2), d.get('train_step_ms'), d.get('train'))"
