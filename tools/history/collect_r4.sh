#!/bin/bash
# Developer tool (GPU box): the round-3 measurement set -> gpurun_out/r4/.  bash tools/collect_r4.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
b() { name=$1; shift; python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"; }
b default
b cfg2_bf16 --dtype bf16 --steps 100 --warmup 20 --no-cpu-baseline
b cfg3 --workload cfg3 --steps 100 --warmup 20 --no-cpu-baseline
b cfg3_bf16 --workload cfg3 --dtype bf16 --steps 100 --warmup 20 --no-cpu-baseline
b cfg5_fp32 --workload cfg5 --steps 30 --warmup 8 --no-cpu-baseline
b cfg5_bf16 --workload cfg5 --dtype bf16 --steps 30 --warmup 8 --no-cpu-baseline
b ref --workload ref --steps 100 --warmup 20 --no-cpu-baseline
b ref_crop --workload ref-crop --steps 100 --warmup 20 --no-cpu-baseline
SCN_EXEC=0 python bench.py --dtype bf16 --steps 100 --warmup 20 --no-cpu-baseline --no-extras > $O/bench_cfg2_bf16_no_executor.json 2>/dev/null; echo "bench bf16 noexec rc=$?"
SCN_EXEC=0 python bench.py --workload cfg3 --dtype bf16 --steps 100 --warmup 20 --no-cpu-baseline --no-extras > $O/bench_cfg3_bf16_no_executor.json 2>/dev/null; echo "bench cfg3 bf16 noexec rc=$?"
SCN_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n2_gloo_one_gpu.json 2> $O/bench_n2.err; echo "bench n2 rc=$?"
SCN_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload cfg3 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n2_gloo_cfg3_bf16.json 2> $O/bench_n2b.err; echo "bench n2 cfg3 rc=$?"
SCN_BENCH_FORCE_DIST=1 SCN_DP_FORCE_BUCKETS=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_rccl_one_rank.json 2> $O/bench_rccl.err; echo "bench rccl-1 rc=$?"
python tools/index_fused_ab.py 30 2>&1 | grep -v amdgpu.ids > $O/index_fused_ab.txt; echo "index ab rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -o cfg2 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_cfg2.log 2>&1; echo "prof cfg2 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2_bf16 -o cfg2_bf16 -- python3 $R/bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_cfg2_bf16.log 2>&1; echo "prof bf16 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg3 -o cfg3 -- python3 $R/bench.py --workload cfg3 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_cfg3.log 2>&1; echo "prof cfg3 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ref -o ref -- python3 $R/bench.py --workload ref --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_ref.log 2>&1; echo "prof ref rc=$?"
rocprofv3 --kernel-trace -d $O/prof_index -o idx -- python3 $R/tools/index_fused_profile.py cfg2 20 > $O/prof_index.log 2>&1; echo "prof index rc=$?"
python3 $R/tools/rocpd_summary.py $O/prof_index/idx_results.db 198 217 > $O/index_build_trace.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_$c.log 2>&1; echo "pmc $c rc=$?"; done
cd $R
python tools/collect_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/traffic.json cfg2 f32
rm -f $O/prof_*/*kernel_trace.csv $O/pmc_*/*.csv $O/prof_index/*.db
find $O -name "*kernel_stats.csv" | head
ls $O
