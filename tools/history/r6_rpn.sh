#!/bin/bash
# round 6: the RPN chain tests + first bench lines of --workload ref-crop-rpn
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_atsize.py tests/test_gpu_parity.py -m gpu -x -q -k "rpn_chain or dense_rpn_stack or rpn_boundary" > $O/pytest_rpn.log 2>&1; rc=$?
tail -15 $O/pytest_rpn.log
[ $rc -ne 0 ] && exit $rc
for dt in f32 bf16; do
  timeout -k 10 300 python bench.py --workload ref-crop-rpn --dtype $dt --steps 20 --warmup 8 --no-cpu-baseline > $O/bench_refcroprpn_$dt.json 2> $O/bench_refcroprpn_$dt.err || { echo "bench $dt failed"; tail -5 $O/bench_refcroprpn_$dt.err; }
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_refcroprpn_$dt.json").read().strip().splitlines()[-1])
    print("$dt", d["ms_per_step"], d["value"], d.get("forward_only_ms"), d["peak_hbm_bytes"]/1e9)
except Exception as e: print("no line", e)
PY
done
