#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r6e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
DT=${1:-f32}
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$DT -o rpn -- python3 $GRAFT_REPO_ROOT/bench.py --workload ref-crop-rpn --dtype $DT --steps 6 --warmup 4 --no-cpu-baseline --no-extras > $O/bench_prof_$DT.json 2> $O/bench_prof_$DT.err || { echo prof failed; tail -5 $O/bench_prof_$DT.err; }
f=$(find $O/prof_$DT -name "*kernel_stats.csv" | head -1)
head -25 "$f"
