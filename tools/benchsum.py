"""one line per bench.py / tools/train_bench.py JSON: value, ms per step, per-kernel-family ms (kernel_ms_per_step)"""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    print(f, round(d["value"], 2), round(d["ms_per_step"], 2))
    rows = []
    for k, v in d.get("kernel_ms_per_step", {}).items():
        if isinstance(v, dict):
            rows.append(f"{k[:26]}={v['total']:.2f}(" + ",".join(f"{x:.2f}" for n, x in v.items() if n != "total") + ")")
        elif v > 0:
            rows.append(f"{k.split('(')[0][:26]}={v:.2f}")
    print("   " + "  ".join(rows))
