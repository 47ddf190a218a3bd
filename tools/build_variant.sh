#!/bin/bash
# Developer tool: build a variant of the library for an A/B on one box (tools/ab_lib.py).
#   tools/build_variant.sh <out.so> <file.hip> "<extra hipcc flags>"   -- recompiles ONE translation unit with the flags and
# links it with the current objects of the others (csrc/build/*.o must be up to date: run build() first).
set -e
OUT=$1; F=$2; FLAGS=$3
R=$(cd $(dirname $0)/.. && pwd); C=$R/sparse_rcnn_amd/csrc
mkdir -p $(dirname $OUT) /tmp/variant
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c $C/$F -o /tmp/variant/$F.o
OBJS=""
for o in $C/build/*.hip.o; do
  if [ "$(basename $o)" == "$F.o" ]; then OBJS="$OBJS /tmp/variant/$F.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o $OUT $OBJS
echo built $OUT
