#!/bin/bash
# round 5, first GPU pass: the whole -m gpu suite, then the bench lines
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5a
python -m pytest tests -m gpu -x -q > gpurun_out/r5a/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r5a/pytest.log
tail -5 gpurun_out/r5a/pytest.log
python bench.py > gpurun_out/r5a/bench_default.json 2> gpurun_out/r5a/bench_default.err; echo "bench rc=$?"
python bench.py --workload cfg3-rpn --no-cpu-baseline > gpurun_out/r5a/bench_cfg3rpn.json 2> gpurun_out/r5a/bench_cfg3rpn.err; echo "rpn rc=$?"
python bench.py --workload cfg3-rpn --dtype bf16 --no-cpu-baseline > gpurun_out/r5a/bench_cfg3rpn_bf16.json 2> gpurun_out/r5a/bench_cfg3rpn_bf16.err; echo "rpn bf16 rc=$?"
python bench.py --workload cfg3 --no-cpu-baseline > gpurun_out/r5a/bench_cfg3.json 2> gpurun_out/r5a/bench_cfg3.err; echo "cfg3 rc=$?"
