#!/bin/bash
# round 6, first GPU pass: per-level launch times of the tile kernels (fp32 / bf16 / weight gradient) and the default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6a; mkdir -p $O
for l in 0 1 2 3; do python tools/ablate_conv.py $l > $O/ablate_conv_$l.txt 2>&1 || echo "ablate_conv $l failed"; done
python tools/ablate_conv_bf16.py > $O/ablate_conv_bf16.txt 2>&1 || echo "ablate bf16 failed"
python tools/ablate_wgrad_direct.py > $O/ablate_wgrad.txt 2>&1 || echo "ablate wgrad failed"
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -3 $O/ablate_conv_0.txt
