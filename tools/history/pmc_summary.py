"""Developer tool: per-kernel sums of a rocprofv3 --pmc counter_collection.csv (averaged over dispatches)."""
import csv, glob, sys, collections
rows = list(csv.DictReader(open((glob.glob(sys.argv[1] + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*counter_collection.csv"))[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "dispatches", len(next(iter(d.values()))))
