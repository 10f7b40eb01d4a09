AB_OFFSETS=1,2,3,4 AB_ROUNDS=6 AB_STEPS=8 python tools/split_ab.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['arm'], round(d['ms_median'],2), round(d['ms_min'],2), d['all'], d['equal_to_one_stream'])"
