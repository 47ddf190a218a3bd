#!/bin/bash
# round 5: do the matrix kernels stall on CUs another stream's kernels hold?  1-rank NCCL process group with 4 overlapped buckets
# (the collective moves nothing; its kernels still occupy CUs) and the plain step, with the matrix grids sized to 256 / 248 / 240 CUs.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r5m; mkdir -p $O
run() { tag=$1; shift; v=$(env "$@" python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-extras $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"); echo "$tag: $v" | tee -a $O/cu_budget.txt; }
for rep in 1 2; do
for B in 256 248 240 224; do
ARGS=""; run "plain step, budget $B" SCN_CU_BUDGET=$B
ARGS="--buckets 4"; run "1-rank NCCL 4 buckets, NCCL_MAX_NCHANNELS=4, budget $B" SCN_BENCH_FORCE_DIST=1 SCN_DP_FORCE_BUCKETS=1 NCCL_MAX_NCHANNELS=4 SCN_CU_BUDGET=$B
done
done
