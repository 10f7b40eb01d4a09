import csv, collections, sys
rows=list(csv.reader(open(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/launches.csv')))
agg=collections.OrderedDict()
for row in rows:
    cls, label, ms, fl, by = row[0], ",".join(row[1:-3]), row[-3], row[-2], row[-1]
    k=(cls,label)
    a=agg.setdefault(k,[0,0.0,float(fl),float(by)])
    a[0]+=1; a[1]+=float(ms)
for (cls,label),a in sorted(agg.items(), key=lambda kv:-kv[1][1])[:int(sys.argv[2]) if len(sys.argv)>2 else 24]:
    ms=a[1]/a[0]
    print(f"{cls[:14]:14s} {label[:50]:50s} n={a[0]:3d} ms={ms:7.3f} TF/s={a[2]/ms/1e9:8.1f} GB/s={a[3]/ms/1e6:8.1f}")
