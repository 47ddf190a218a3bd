#!/bin/bash
# A/B of one environment switch on bench.py: tools/ab_env.sh VAR "bench args" [rounds] -> ms_per_step of alternating runs
# (VAR unset, VAR=0, VAR unset, ...).  One box, same process conditions; prints every run.
VAR=$1; ARGS=$2; N=${3:-3}
for i in $(seq 1 $N); do
  for v in on off; do
    if [ $v = off ]; then export $VAR=0; else unset $VAR; fi
    timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline --no-extras > /tmp/ab.json 2>/tmp/ab.err || { echo "run failed"; tail -3 /tmp/ab.err; exit 1; }
    python -c "import json; d=json.load(open('/tmp/ab.json')); print('$VAR', '$v', '$ARGS', round(d['ms_per_step'], 3))"
  done
done
