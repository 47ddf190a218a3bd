#!/bin/bash
# round 6: (row bin, mask) tiles at level 0 for the bf16 kernels -- full GPU suite, then the A/B in the steps
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6j; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
b() { name=$1; shift; timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-extras > $O/bench_$name.json 2> $O/bench_$name.err || echo "bench $name failed";
python - <<PY
import json
d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
print("$name", round(d["ms_per_step"],3), d["roofline"].get("frac"), d["roofline"].get("achieved"))
PY
}
for r in 1 2; do
SCN_TB_NO_BINS=1 b cfg5_bf16_nobins_$r --workload cfg5 --dtype bf16 --steps 30 --warmup 8
b cfg5_bf16_bins_$r --workload cfg5 --dtype bf16 --steps 30 --warmup 8
SCN_TB_NO_BINS=1 b cfg2_bf16_nobins_$r --dtype bf16 --steps 60 --warmup 15
b cfg2_bf16_bins_$r --dtype bf16 --steps 60 --warmup 15
done
b cfg2_f32 --steps 60 --warmup 15
