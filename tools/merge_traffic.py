"""profiles/r5_traffic.json: the per-workload files of tools/collect_traffic.py merged into one, an entry per (workload, dtype).
    python tools/merge_traffic.py out.json in1.json in2.json ..."""
import json, sys
ents = [json.load(open(p)) for p in sys.argv[2:]]
json.dump(dict(note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --workload W --dtype D --steps 2 "
                    "--warmup 1`; FETCH_SIZE x2 (gfx950, 16-B/lane reads); includes Infinity-Cache hits; bytes per launch, "
                    "averaged over the launches of a kernel name", entries=ents), open(sys.argv[1], "w"), indent=1)
print([(e["workload"], e["dtype"], sorted(e["kernels"])[:4]) for e in ents])
