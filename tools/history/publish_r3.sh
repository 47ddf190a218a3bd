#!/bin/bash
# Developer tool (build container): copy what tools/collect_r3.sh left in gpurun_out/r3/ into profiles/ (tracked).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/gpurun_out/r3
P=$R/profiles
for f in $S/bench_*.json; do cp $f $P/r3_$(basename $f); done
[ -f $S/ablate_conv_tail.txt ] && cp $S/ablate_conv_tail.txt $P/r3_ablate_conv_tail_final_weights.txt
cp $S/traffic.json $P/r3_traffic.json
cp $S/prof_cfg2/cfg2_kernel_stats.csv $P/r3_kernel_stats.csv
cp $S/prof_cfg2_bf16/cfg2_bf16_kernel_stats.csv $P/r3_kernel_stats_cfg2_bf16.csv
cp $S/prof_cfg3/cfg3_kernel_stats.csv $P/r3_kernel_stats_cfg3.csv
cp $S/prof_ref/ref_kernel_stats.csv $P/r3_kernel_stats_ref.csv
ls $P | grep r3_ | wc -l
