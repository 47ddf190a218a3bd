#!/bin/bash
# round 5 A/B: host-side order of a detection + mask step (bf16 storage).  Alternating runs on one box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r5g; mkdir -p $O
run() { tag=$1; shift; v=$(env "$@" python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-extras $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['per_rank'][0]['n_roi_rows'])"); echo "$WL $tag: $v" | tee -a $O/ab_host.txt; }
for rep in 1 2; do
WL="cfg3 bf16"; ARGS="--workload cfg3 --dtype bf16"
run "base (prefetch thread first)" SCN_LATE_PREFETCH=0
run "late prefetch" SCN_LATE_PREFETCH=1
run "late prefetch + inline backward" SCN_LATE_PREFETCH=1 SCN_BACKWARD_INLINE=1
WL="cfg3-rpn bf16"; ARGS="--workload cfg3-rpn --dtype bf16"
run "rpn after decoder" SCN_RPN_EARLY=0
run "rpn between encoder and decoder" SCN_RPN_EARLY=1
run "rpn early + inline backward" SCN_RPN_EARLY=1 SCN_BACKWARD_INLINE=1
WL="cfg2 bf16"; ARGS="--dtype bf16"
run "base" SCN_LATE_PREFETCH=0
run "late prefetch" SCN_LATE_PREFETCH=1
run "late + inline backward" SCN_LATE_PREFETCH=1 SCN_BACKWARD_INLINE=1
done
WL="cfg2 f32"; ARGS=""
run "late prefetch" SCN_LATE_PREFETCH=1
run "base" SCN_LATE_PREFETCH=0
run "late + inline backward" SCN_LATE_PREFETCH=1 SCN_BACKWARD_INLINE=1
