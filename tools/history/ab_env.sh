#!/bin/bash
# A/B of one environment switch on bench.py, alternating runs on one box:
#   tools/ab_env.sh VAR A_VALUE B_VALUE "bench args" [rounds]        ("-" = variable unset)
# prints ms_per_step of every run.
VAR=$1; A=$2; B=$3; ARGS=$4; N=${5:-3}
for i in $(seq 1 $N); do
  for v in "$A" "$B"; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline --no-extras > /tmp/ab.json 2>/tmp/ab.err || { echo "run failed"; tail -3 /tmp/ab.err; exit 1; }
    python -c "import json; d=json.load(open('/tmp/ab.json')); print('$VAR=$v', '$ARGS', round(d['ms_per_step'], 3))"
  done
done
