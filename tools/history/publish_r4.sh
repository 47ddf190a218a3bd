#!/bin/bash
# Developer tool (build container): copy what tools/collect_r4.sh left in gpurun_out/r4/ into profiles/ (tracked).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/gpurun_out/r4
P=$R/profiles
for f in $S/bench_*.json; do cp $f $P/r4_$(basename $f); done
cp $S/index_fused_ab.txt $P/r4_index_fused_ab.txt
cp $S/index_build_trace.txt $P/r4_index_build_trace.txt
cp $S/traffic.json $P/r4_traffic.json
cp $S/prof_cfg2/cfg2_kernel_stats.csv $P/r4_kernel_stats.csv
cp $S/prof_cfg2_bf16/cfg2_bf16_kernel_stats.csv $P/r4_kernel_stats_cfg2_bf16.csv
cp $S/prof_cfg3/cfg3_kernel_stats.csv $P/r4_kernel_stats_cfg3.csv
cp $S/prof_ref/ref_kernel_stats.csv $P/r4_kernel_stats_ref.csv
ls $P | grep r4_ | wc -l
