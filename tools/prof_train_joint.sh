# rocprofv3 --kernel-trace --stats of the joint-target training step (the shipped config): what the splat / pyramid-loss side costs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_joint
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_joint -- python tools/train_bench.py --target joint --height 448 --steps 4 --warmup 2 > gpurun_out/prof_joint.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_joint/*/*kernel_stats.csv')[0]
open('gpurun_out/prof_joint_kernel_stats.csv', 'w').write(open(f).read())
rows = list(csv.DictReader(open(f)))
print('total kernel ms/step', sum(float(r['TotalDurationNs']) for r in rows) / 6e6)
for r in rows:
    n = r['Name']
    if not any(k in n for k in ('conv', 'gnbwd', 'lc_', 'la_', 'layernorm', 'resblock', 'fa_bwd', 'flash', 'wgrad', 'gn_finalize', 'grad_', 'affine_silu', 'block_mlp', 'adam', 'wt_transpose')):
        if float(r['TotalDurationNs']) / 6e6 > 0.05:
            print(f"{n[:100]:100s} calls/step={int(r['Calls'])/6:7.1f} ms/step={float(r['TotalDurationNs'])/6e6:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
