cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_unet_gpu.py -m gpu -q -k "forward_vs_engine" 2>&1 | tail -2
rm -rf gpurun_out/prof_tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/prof_tmp.log 2>&1
grep '"metric"' gpurun_out/prof_tmp.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps/s', d['value'], 'ms', d['ms_per_step'])"
python - <<'PY'
import csv, glob
f=glob.glob('gpurun_out/prof_tmp/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
