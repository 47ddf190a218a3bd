set -e
cd /root/repo
for i in 1 2 3; do
 for pf in stream 0; do
  for dt in f32 bf16; do
   echo "== prefetch=$pf dtype=$dt rep=$i" >> gpurun_out/ab_roi.log
   SCN_ROI_PREFETCH=$pf python bench.py --workload cfg3 --dtype $dt --no-extras --steps 30 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])" >> gpurun_out/ab_roi.log
  done
 done
done
cat gpurun_out/ab_roi.log
