#!/bin/bash
# Developer tool (GPU box): the round-2 measurement set -> gpurun_out/r2/.  bash tools/collect_r2.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2
mkdir -p $O
cd $R
b() { name=$1; shift; python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"; }
b cfg2 --dropin
b cfg2_all_kernels --steps 20 --warmup 5 --profile-all --no-cpu-baseline --no-extras
b cfg2_bf16 --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline
b cfg3 --workload cfg3 --steps 20 --warmup 5 --no-cpu-baseline
b cfg3_bf16 --workload cfg3 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
b cfg5_fp32 --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline
b cfg5_bf16 --workload cfg5 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline
SCN_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n2_gloo_one_gpu.json 2> $O/bench_n2.err; echo "bench n2 rc=$?"
for l in 0 1 2 3; do python tools/ablate_conv.py $l; done 2>&1 | grep -v amdgpu.ids > $O/ablate_conv.txt
python tools/ablate_conv_bf16.py 2>&1 | grep -v amdgpu.ids > $O/ablate_conv_bf16.txt
python tools/ablate_wgrad_bf16.py 2>&1 | grep -v amdgpu.ids > $O/ablate_wgrad_bf16.txt
python tools/ablate_gemm.py 2>&1 | grep -v amdgpu.ids > $O/ablate_gemm.txt
python tools/bench_roi_crop.py 2>&1 | grep -v amdgpu.ids > $O/roi_crop.txt
python tools/diag_bf16_layers.py 2>&1 | grep -v amdgpu.ids > $O/diag_bf16_layers.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -o cfg2 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_cfg2.log 2>&1; echo "prof cfg2 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2_bf16 -o cfg2_bf16 -- python3 $R/bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_cfg2_bf16.log 2>&1; echo "prof bf16 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg3 -o cfg3 -- python3 $R/bench.py --workload cfg3 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_cfg3.log 2>&1; echo "prof cfg3 rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_$c.log 2>&1; echo "pmc $c rc=$?"; done
cd $R
python tools/collect_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/traffic.json cfg2 f32
rm -f $O/prof_*/*kernel_trace.csv $O/pmc_*/*.csv
ls $O
