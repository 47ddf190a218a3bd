#!/bin/bash
# Developer tool (build container): copy what tools/collect_r5.sh left in gpurun_out/r5/ into profiles/ (tracked).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/gpurun_out/r5
P=$R/profiles
for f in $S/bench_*.json; do cp $f $P/r5_$(basename $f); done
cp $S/index_fused_ab.txt $P/r5_index_fused_ab.txt
[ -f $S/index_build_trace.txt ] && cp $S/index_build_trace.txt $P/r5_index_build_trace.txt
cp $S/traffic.json $P/r5_traffic.json
cp $S/long_run.txt $P/r5_long_run.txt
cp $S/sq_counters_fp32.txt $P/r5_sq_counters_fp32.txt
cp $S/sq_counters_bf16.txt $P/r5_sq_counters_bf16.txt
cp $S/prof_cfg2/cfg2_kernel_stats.csv $P/r5_kernel_stats.csv
cp $S/prof_cfg2_bf16/cfg2_bf16_kernel_stats.csv $P/r5_kernel_stats_cfg2_bf16.csv
cp $S/prof_cfg3_bf16/cfg3_bf16_kernel_stats.csv $P/r5_kernel_stats_cfg3_bf16.csv
cp $S/prof_cfg3rpn/cfg3rpn_kernel_stats.csv $P/r5_kernel_stats_cfg3rpn.csv
cp $S/prof_cfg3rpn_bf16/cfg3rpn_bf16_kernel_stats.csv $P/r5_kernel_stats_cfg3rpn_bf16.csv
cp $S/prof_cfg5_bf16/cfg5_bf16_kernel_stats.csv $P/r5_kernel_stats_cfg5_bf16.csv
ls $P | grep r5_ | wc -l
