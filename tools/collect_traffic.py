"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) of
`bench.py` into profiles/<name>_traffic.json: HBM-side bytes per launch of each hot kernel.

    python tools/collect_traffic.py <fetch_dir> <write_dir> <out.json> [workload] [dtype]

gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports half the bytes of 16-byte-per-lane reads -> x2;
WRITE_SIZE is exact for 16-byte stores.  Counters are in KB and include Infinity-Cache hits."""
import csv, glob, json, sys, collections

def load(d, name):
    rows = list(csv.DictReader(open((glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0])))
    agg = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            agg[k].append(float(r["Counter_Value"]))
    return agg

f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    if not k.startswith("k_"):
        continue
    fb = sum(f.get(k, [0])) / max(1, len(f.get(k, [0]))) * 1024.0
    wb = sum(w.get(k, [0])) / max(1, len(w.get(k, [0]))) * 1024.0
    out[k] = dict(launches=len(f.get(k, [])), fetch_bytes_raw=fb, fetch_bytes_corrected=2 * fb, write_bytes=wb,
                  hbm_bytes_per_launch=2 * fb + wb)
json.dump(dict(workload=sys.argv[4] if len(sys.argv) > 4 else "cfg2", dtype=sys.argv[5] if len(sys.argv) > 5 else "f32",
               note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1`; "
                    "FETCH_SIZE x2 (gfx950, 16-B/lane reads); includes Infinity-Cache hits", kernels=out),
          open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out.items()}))
