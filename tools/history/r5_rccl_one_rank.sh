#!/bin/bash
# round 5: what the RCCL calls cost a step on ONE GPU (a 1-rank process group: the all-reduce moves nothing, its launches and the
# bucket copies remain) -- overlapped buckets vs one all-reduce after backward vs no process group.  Alternating runs.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r5m; mkdir -p $O
run() { tag=$1; shift; v=$(env "$@" python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-extras $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['per_rank'][0]['allreduce_ms_exposed'])"); echo "$tag: $v" | tee -a $O/rccl_one_rank.txt; }
for rep in 1 2 3; do
ARGS=""; run "no process group" X=1
ARGS="--buckets 4"; run "1-rank NCCL, 4 overlapped buckets" SCN_BENCH_FORCE_DIST=1 SCN_DP_FORCE_BUCKETS=1
ARGS="--buckets 1"; run "1-rank NCCL, 1 bucket from the last hook" SCN_BENCH_FORCE_DIST=1 SCN_DP_FORCE_BUCKETS=1
ARGS="--buckets 0"; run "1-rank NCCL, one all-reduce after backward" SCN_BENCH_FORCE_DIST=1 SCN_DP_FORCE_BUCKETS=1
ARGS="--buckets 4"; run "1-rank NCCL, 4 buckets, NCCL_MAX_NCHANNELS=4" SCN_BENCH_FORCE_DIST=1 SCN_DP_FORCE_BUCKETS=1 NCCL_MAX_NCHANNELS=4
done
