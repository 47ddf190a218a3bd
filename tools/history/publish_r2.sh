#!/bin/bash
# Developer tool (build container): copy what tools/collect_r2.sh left in gpurun_out/r2/ into profiles/ (tracked).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/gpurun_out/r2
P=$R/profiles
for f in $S/bench_*.json; do cp $f $P/r2_$(basename $f); done
for t in ablate_conv ablate_conv_bf16 ablate_gemm ablate_wgrad_bf16 roi_crop diag_bf16_layers; do [ -f $S/$t.txt ] && cp $S/$t.txt $P/r2_$t.txt; done
cp $S/traffic.json $P/r2_traffic.json
cp $S/prof_cfg2/cfg2_kernel_stats.csv $P/r2_kernel_stats.csv
cp $S/prof_cfg2_bf16/cfg2_bf16_kernel_stats.csv $P/r2_kernel_stats_cfg2_bf16.csv
cp $S/prof_cfg3/cfg3_kernel_stats.csv $P/r2_kernel_stats_cfg3.csv
ls $P | grep r2_ | wc -l
