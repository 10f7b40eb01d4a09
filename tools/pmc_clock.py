"""Effective shader clock and MFMA-busy fraction per kernel from a `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES
SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES` pass (tools/pmc.sh): clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration."""
import collections, csv, glob, json, sys
cc = sorted(glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES") + "/*/*counter_collection.csv"))[-1]
agg = collections.defaultdict(lambda: [0.0, 0, 0, 0.0])
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"].split("(")[0][:60]
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        agg[k][0] += float(r["Counter_Value"]) / 8
        agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[k][2] += 1
    elif r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        agg[k][3] += float(r["Counter_Value"])
out = {}
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if v[1] / max(v[2], 1) < 20e3:
        continue
    out[k] = {"launches": v[2], "avg_us": round(v[1] / v[2] / 1e3, 1), "clock_GHz": round(v[0] / v[1], 3), "mfma_busy_frac": round(v[3] / (v[0] * 1024), 3)}
    print(f"{k:60s} {out[k]}")
json.dump({"source": cc, "note": "clock = GRBM_GUI_ACTIVE / 8 / (End - Start); nominal 2.4 GHz: the dense-bf16 peak a kernel can reach is 2.5 PF x clock / 2.4", "kernels": out},
          open("gpurun_out/effective_clock.json", "w"), indent=1)
