# SQ-side counters per kernel (why a kernel's MFMA pipe idles): separate passes of <= 8 SQ counters each, --kernel-trace only
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1 || true
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rm -rf gpurun_out/sq_$i
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/sq_$i -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --train-steps 0 > gpurun_out/sq_$i.log 2>&1
  echo "== pass $i rc=$? : $(ls gpurun_out/sq_$i/*/ 2>/dev/null | tr '\n' ' ')"
done
python - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
for i in (1, 2, 3):
    for f in glob.glob(f'gpurun_out/sq_{i}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:70].strip()
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0))
out = {}
for k, v in rows[:16]:
    n = max(cnt[k].values())
    out[k] = {"launches": n, **{c: round(val / cnt[k][c], 1) for c, val in sorted(v.items())}}
json.dump(out, open('gpurun_out/sq_summary.json', 'w'), indent=1)
for k, v in out.items():
    print(k); print('   ', {c: x for c, x in v.items()})
PY
