#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6b; mkdir -p $O
for l in 0 1; do for g in 0 8; do
  echo "--- level $l SCN_TS_PROG=$g ---" >> $O/prog_timeline.txt
  SCN_TS_PROG=$g TL_LAUNCHES=1500 SCN_MI355X_LIB=$PWD/tools/ab/libscn_tl.so timeout -k 10 120 python tools/ts_timeline.py $l 2>&1 | head -8 >> $O/prog_timeline.txt || true
done; done
cat $O/prog_timeline.txt
