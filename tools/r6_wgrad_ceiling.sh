#!/bin/bash
# round 6 (c): ceiling of a one-gather-per-rule weight gradient, measured on compiled variants of k_wgrad_direct
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6h; mkdir -p $O
for v in base wdexp1 wdexp2 base; do
  echo "=== $v ===" >> $O/wgrad_ceiling.txt
  if [ $v == base ]; then timeout -k 10 200 python tools/ablate_wgrad_direct.py 2>&1 | grep "subm" >> $O/wgrad_ceiling.txt
  else SCN_MI355X_LIB=$PWD/tools/ab/libscn_$v.so timeout -k 10 200 python tools/ablate_wgrad_direct.py 2>&1 | grep "subm" >> $O/wgrad_ceiling.txt; fi
done
cat $O/wgrad_ceiling.txt
