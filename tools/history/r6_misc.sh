#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6f; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_dense_twin_golden.py tests/test_gpu_parity.py -m gpu -x -q -k "dense_twin or references_dense or dense_rpn_stack or roi" > $O/pytest.log 2>&1; rc=$?
tail -6 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --workload cfg2-bn --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_cfg2bn.json 2> $O/bench_cfg2bn.err || { echo "bench bn failed"; tail -5 $O/bench_cfg2bn.err; }
python - <<PY
import json
d=json.loads(open("$O/bench_cfg2bn.json").read().strip().splitlines()[-1])
print("cfg2-bn", d["ms_per_step"], d["value"], d.get("forward_only_ms"))
PY
