# SQ / TCC counters per kernel for an arbitrary python script: bash tools/pmc_any.sh <tag> <script.py> [args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_FLAT GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1)); rm -rf gpurun_out/pa_${TAG}_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pa_${TAG}_$i -- python "$@" > gpurun_out/pa_${TAG}_$i.log 2>&1
  echo "== pass $i rc=$?"
done
python - $TAG <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob(f'gpurun_out/pa_{tag}_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:70].strip()
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
out = {}
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0))[:12]:
    out[k] = {"launches": max(cnt[k].values()), **{c: round(val / cnt[k][c], 1) for c, val in sorted(v.items())}}
json.dump(out, open(f'gpurun_out/pa_{tag}_summary.json', 'w'), indent=1)
for k, v in out.items():
    cyc = v.get('GRBM_GUI_ACTIVE', 8) / 8
    print(k, f"cycles/launch {cyc:.0f}")
    print('   ', {c: (x if c in ('launches',) else round(x / cyc, 2)) for c, x in v.items()})
PY
