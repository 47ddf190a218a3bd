#!/bin/bash
# round 6: progressive staging probe + in-step A/B (SCN_TS_PROG) + the in-kernel timeline
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6b; mkdir -p $O
timeout -k 10 300 python tools/r6_prog_probe.py > $O/prog_probe.txt 2>&1 || { echo "probe failed"; tail -20 $O/prog_probe.txt; exit 1; }
cat $O/prog_probe.txt
for v in 0 8 0 8 4 12; do
  SCN_TS_PROG=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 60 --warmup 15 > $O/bench_prog$v.json 2>> $O/bench.err || echo "bench $v failed"
  python - <<PY
import json
d=json.loads(open("$O/bench_prog$v.json").read().strip().splitlines()[-1])
print("PROG=$v", d["ms_per_step"], d["roofline"]["frac"])
PY
done
