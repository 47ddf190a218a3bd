#!/bin/bash
for i in 1 2; do
for lib in default tools/ab/ts_nw12.so; do
for side in 0 1; do
  if [ $lib = default ]; then unset SCN_MI355X_LIB; else export SCN_MI355X_LIB=$PWD/$lib; fi
  export SCN_EXEC_SIDE=$side
  timeout -k 10 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-extras > /tmp/ab.json 2>/tmp/ab.err || { echo failed; tail -3 /tmp/ab.err; }
  python -c "import json; d=json.load(open('/tmp/ab.json')); print('$lib side=$side', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_us'],1))"
done; done; done
