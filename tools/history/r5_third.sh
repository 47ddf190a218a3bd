#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5f
mkdir -p $O
python bench.py --workload cfg3-rpn --no-cpu-baseline > $O/bench_cfg3rpn.json 2> $O/bench_cfg3rpn.err; echo "rpn rc=$?"
python bench.py --workload cfg3-rpn --dtype bf16 --no-cpu-baseline > $O/bench_cfg3rpn_bf16.json 2> $O/bench_cfg3rpn_bf16.err; echo "rpn bf16 rc=$?"
cd /tmp && export TMPDIR=/tmp
for w in cfg3 cfg3-rpn; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${w}_bf16 -o t -- python3 $R/bench.py --workload $w --dtype bf16 --steps 20 --warmup 8 --no-cpu-baseline --no-extras > $O/prof_${w}_bf16.log 2>&1; echo "prof $w rc=$?"
python3 $R/tools/stream_gaps.py $O/prof_${w}_bf16 > $O/gaps_${w}_bf16.txt 2>&1
done
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
ls $O $O/prof_cfg3_bf16
