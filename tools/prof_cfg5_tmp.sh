R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/p5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16 -o cfg5_bf16 -- python3 $R/bench.py --workload cfg5 --dtype bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $O/bf16.log 2>&1 && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/f32 -o cfg5_f32 -- python3 $R/bench.py --workload cfg5 --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $O/f32.log 2>&1
rm -f $O/*/*kernel_trace.csv
ls $O/*
