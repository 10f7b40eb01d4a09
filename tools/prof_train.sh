# rocprofv3 --kernel-trace --stats of the training step alone (tools/train_bench.py): per-kernel totals incl. the small launches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_train
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python tools/train_bench.py --steps 4 --warmup 2 > gpurun_out/prof_train.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_train/*/*kernel_stats.csv')[0]
open('gpurun_out/prof_train_kernel_stats.csv', 'w').write(open(f).read())
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot / 1e6, 'over 6 steps (2 warm-up)')
for r in rows[:60]:
    print(f"{r['Name'][:80]:80s} calls/step={int(r['Calls'])/6:7.1f} ms/step={float(r['TotalDurationNs'])/6e6:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
