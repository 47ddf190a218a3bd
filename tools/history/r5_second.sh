#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5c
python -m pytest tests -m gpu -q -k "rpn or dense_rpn" > gpurun_out/r5c/pytest_rpn.log 2>&1; echo "rpn tests rc=$?"; tail -3 gpurun_out/r5c/pytest_rpn.log
python -m pytest tests -m gpu -q --deselect tests/test_gpu_atsize.py::test_cfg3_rpn_chain_vs_oracle_at_150k > gpurun_out/r5c/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r5c/pytest.log
python bench.py --workload cfg3-rpn --no-cpu-baseline > gpurun_out/r5c/bench_cfg3rpn.json 2> gpurun_out/r5c/bench_cfg3rpn.err; echo "rpn rc=$?"
python bench.py --workload cfg3-rpn --dtype bf16 --no-cpu-baseline > gpurun_out/r5c/bench_cfg3rpn_bf16.json 2> gpurun_out/r5c/bench_cfg3rpn_bf16.err; echo "rpn bf16 rc=$?"
