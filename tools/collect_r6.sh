#!/bin/bash
# Developer tool (GPU box): the round-6 measurement set -> gpurun_out/r6/.  bash tools/collect_r6.sh [part]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
PART=${1:-all}
b() { name=$1; shift; python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"; }
if [ $PART = all ] || [ $PART = bench ]; then
b default
b cfg2_bf16 --dtype bf16 --steps 100 --warmup 20 --no-cpu-baseline
b cfg3 --workload cfg3 --steps 100 --warmup 20 --no-cpu-baseline
b cfg3_bf16 --workload cfg3 --dtype bf16 --steps 100 --warmup 20 --no-cpu-baseline
b cfg3rpn --workload cfg3-rpn --steps 100 --warmup 20 --no-cpu-baseline
b cfg3rpn_bf16 --workload cfg3-rpn --dtype bf16 --steps 100 --warmup 20 --no-cpu-baseline
b cfg5_fp32 --workload cfg5 --steps 30 --warmup 8 --no-cpu-baseline
b cfg5_bf16 --workload cfg5 --dtype bf16 --steps 30 --warmup 8 --no-cpu-baseline
b ref --workload ref --steps 100 --warmup 20 --no-cpu-baseline
b ref_crop --workload ref-crop --steps 100 --warmup 20 --no-cpu-baseline
b ref_crop_rpn --workload ref-crop-rpn --steps 40 --warmup 10 --no-cpu-baseline
b ref_crop_rpn_bf16 --workload ref-crop-rpn --dtype bf16 --steps 40 --warmup 10 --no-cpu-baseline
b cfg2_bn --workload cfg2-bn --steps 60 --warmup 15 --no-cpu-baseline
SCN_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n2_gloo_one_gpu.json 2> $O/bench_n2.err; echo "bench n2 rc=$?"
SCN_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload cfg3-rpn --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n2_gloo_cfg3rpn_bf16.json 2> $O/bench_n2b.err; echo "bench n2 cfg3-rpn rc=$?"
SCN_BENCH_FORCE_DIST=1 SCN_DP_FORCE_BUCKETS=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_rccl_one_rank.json 2> $O/bench_rccl.err; echo "bench rccl-1 rc=$?"
python tools/index_fused_ab.py 30 2>&1 | grep -v amdgpu.ids > $O/index_fused_ab.txt; echo "index ab rc=$?"
python tools/long_run.py 400 2>&1 | grep -v amdgpu.ids > $O/long_run.txt; echo "long run rc=$?"
python tools/long_run.py 400 bf16 2>&1 | grep -v amdgpu.ids >> $O/long_run.txt; echo "long run bf16 rc=$?"
fi
cd /tmp && export TMPDIR=/tmp
if [ $PART = all ] || [ $PART = prof ]; then
p() { name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python3 $R/bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_$name.log 2>&1; echo "prof $name rc=$?"; }
p cfg2
p cfg2_bf16 --dtype bf16
p cfg3_bf16 --workload cfg3 --dtype bf16
p cfg3rpn --workload cfg3-rpn
p cfg3rpn_bf16 --workload cfg3-rpn --dtype bf16
p cfg5_bf16 --workload cfg5 --dtype bf16
p ref_crop_rpn --workload ref-crop-rpn
p ref_crop_rpn_bf16 --workload ref-crop-rpn --dtype bf16
rocprofv3 --kernel-trace -d $O/prof_index -o idx -- python3 $R/tools/index_fused_profile.py cfg2 20 > $O/prof_index.log 2>&1; echo "prof index rc=$?"
python3 $R/tools/rocpd_summary.py $O/prof_index/idx_results.db 198 217 > $O/index_build_trace.txt 2>&1
fi
if [ $PART = all ] || [ $PART = pmc ]; then
t() { name=$1; wl=$2; dt=$3; shift 3
  for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d $O/pmc_${name}_$c -o pmc -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_${name}_$c.log 2>&1; echo "pmc $name $c rc=$?"; done
  python3 $R/tools/collect_traffic.py $O/pmc_${name}_FETCH_SIZE $O/pmc_${name}_WRITE_SIZE $O/traffic_$name.json $wl $dt; }
t cfg2 cfg2 f32
t cfg2_bf16 cfg2 bf16 --dtype bf16
t cfg3_bf16 cfg3 bf16 --workload cfg3 --dtype bf16
t cfg3rpn_bf16 cfg3-rpn bf16 --workload cfg3-rpn --dtype bf16
t cfg5_bf16 cfg5 bf16 --workload cfg5 --dtype bf16
t cfg5 cfg5 f32 --workload cfg5
t ref_crop_rpn ref-crop-rpn f32 --workload ref-crop-rpn
t ref_crop_rpn_bf16 ref-crop-rpn bf16 --workload ref-crop-rpn --dtype bf16
python3 $R/tools/merge_traffic.py $O/traffic.json $O/traffic_cfg2.json $O/traffic_cfg2_bf16.json $O/traffic_cfg3_bf16.json $O/traffic_cfg3rpn_bf16.json $O/traffic_cfg5_bf16.json $O/traffic_cfg5.json $O/traffic_ref_crop_rpn.json $O/traffic_ref_crop_rpn_bf16.json
fi
if [ $PART = all ] || [ $PART = sq ]; then
cd $R; bash tools/collect_sq.sh > $O/sq_collect.log 2>&1; python3 tools/sq_report.py gpurun_out/sq > $O/sq_counters_fp32.txt 2>&1; echo "sq rc=$?"
rm -rf gpurun_out/sq
bash tools/collect_sq.sh --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-extras >> $O/sq_collect.log 2>&1; python3 tools/sq_report.py gpurun_out/sq > $O/sq_counters_bf16.txt 2>&1; echo "sq bf16 rc=$?"
rm -rf gpurun_out/sq
fi
cd $R
rm -rf $O/prof_*/*kernel_trace.csv $O/pmc_*/ $O/prof_index
find $O -name "*agent_info.csv" -delete
find $O -name "*kernel_stats.csv" | head -20
ls $O
