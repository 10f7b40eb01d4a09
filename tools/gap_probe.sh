cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gap
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap -- python tools/gap_probe.py > gpurun_out/gap.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/gap/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 2:]                      # steady state: the second half of the run
ex = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]
gaps = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(rows, rows[1:])]
gaps_s = sorted(gaps)
n = len(gaps)
print(f"kernels {len(rows)}  exec mean {sum(ex)/len(ex)/1e3:.2f} us (median {sorted(ex)[len(ex)//2]/1e3:.2f})  "
      f"gap mean {sum(gaps)/n/1e3:.2f} us  median {gaps_s[n//2]/1e3:.2f}  p10 {gaps_s[n//10]/1e3:.2f}  p90 {gaps_s[9*n//10]/1e3:.2f}")
span = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
print(f"span {span/1e6:.2f} ms  exec total {sum(ex)/1e6:.2f} ms  idle {100*(span-sum(ex))/span:.1f} %")
import collections
by = collections.Counter()
for r, e in zip(rows, ex): by[r['Kernel_Name'][:50]] += e
for k, v in by.most_common(8): print(f"  {k:50s} {v/1e6:.2f} ms")
bg = collections.defaultdict(lambda: [0, 0])
for r, e in zip(rows, ex):
    if 'conv_igemm_kernel<3' in r['Kernel_Name'] or 'pingpong' in r['Kernel_Name']:
        k = (r['Kernel_Name'][:44], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Y', ''))
        bg[k][0] += 1; bg[k][1] += e
for k, (n, t) in sorted(bg.items(), key=lambda kv: -kv[1][1])[:12]: print(f"  {k} n={n} avg={t/n/1e3:.1f} us total={t/1e6:.2f} ms")
PY
rm -rf gpurun_out/gap
