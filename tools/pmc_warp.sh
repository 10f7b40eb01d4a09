# PMC passes for the warp kernels (one counter group per run, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf gpurun_out/pmcw_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmcw_$tag -- python tools/warp_bench.py > gpurun_out/pmcw_$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections, json
out = {}
for tag in ("FETCH_SIZE", "WRITE_SIZE", "SQ_LDS_BANK_CONFLICT"):
    f = glob.glob(f"gpurun_out/pmcw_{tag}/*/*counter_collection.csv")
    if not f: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        for c, val in v.items():
            out.setdefault(k, {})[c + "_per_launch"] = val / cnt[(k, c)]
for k, v in out.items():
    if "FETCH_SIZE_per_launch" in v: v["fetch_MB(x2 gfx950 correction)"] = v["FETCH_SIZE_per_launch"] * 1024 * 2 / 1e6
    if "WRITE_SIZE_per_launch" in v: v["write_MB"] = v["WRITE_SIZE_per_launch"] * 1024 / 1e6
json.dump(out, open("gpurun_out/pmc_warp_summary.json", "w"), indent=1)
for k, v in out.items():
    if "warp" in k or "splat" in k: print(k, {a: round(b, 1) for a, b in v.items()})
PY
