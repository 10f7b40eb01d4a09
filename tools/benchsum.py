import json,sys
for f in sys.argv[1:]:
    d=json.load(open(f))
    print(f, round(d["value"],2), round(d["ms_per_step"],2))
    print("   "+"  ".join(f"{k.split('(')[0][:28]}={v:.2f}" for k,v in d["kernel_ms_per_step"].items() if v>0))
