# rocprofv3 --kernel-trace --stats of the denoise leg alone (bench.py --train-steps 0): the per-kernel averages the
# `roofline` object of bench.py must agree with
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_infer
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_infer -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --train-steps 0 > gpurun_out/prof_infer.log 2>&1
grep '"metric"' gpurun_out/prof_infer.log > gpurun_out/prof_infer_bench.json
python - <<'PY'
import csv, glob, json
f = glob.glob('gpurun_out/prof_infer/*/*kernel_stats.csv')[0]
open('gpurun_out/prof_infer_kernel_stats.csv', 'w').write(open(f).read())
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
d = json.loads(open('gpurun_out/prof_infer_bench.json').read())
print('denoise steps/s', d['value'], 'roofline avg_launch_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])
PY
