set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
python bench.py --workload cfg3-rpn --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_cfg3rpn.json 2> $O/bench_cfg3rpn.err; echo rc=$?
python bench.py --workload cfg3-rpn --dtype bf16 --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_cfg3rpn_bf16.json 2> $O/bench_cfg3rpn_bf16.err; echo rc=$?
python bench.py --workload cfg3 --dtype bf16 --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_cfg3_bf16.json 2> $O/bench_cfg3_bf16.err; echo rc=$?
cd /tmp && export TMPDIR=/tmp
for name in cfg3rpn cfg3rpn_bf16; do
  extra=""; [ $name = cfg3rpn_bf16 ] && extra="--dtype bf16"
  rm -rf $O/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python3 $R/bench.py --workload cfg3-rpn $extra --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_$name.log 2>&1; echo "prof $name rc=$?"
done
cd $R; rm -rf $O/prof_*/*kernel_trace.csv; find $O -name "*agent_info.csv" -delete
