# rocprofv3 --kernel-trace --stats of the default bench.py command (denoise leg + train leg), summary -> gpurun_out/prof_final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_final
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_final.log 2>&1
grep '"metric"' gpurun_out/prof_final.log > gpurun_out/prof_final_bench.json
python - <<'PY'
import csv, glob, json
f = glob.glob('gpurun_out/prof_final/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
open('gpurun_out/prof_final_kernel_stats.csv', 'w').write(open(f).read())
for r in rows[:28]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
d = json.loads(open('gpurun_out/prof_final_bench.json').read())
print('denoise steps/s', d['value'], 'roofline avg_launch_ms', d['roofline']['avg_launch_ms'], 'train', d.get('train', {}).get('value'))
PY
