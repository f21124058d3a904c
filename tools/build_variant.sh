#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." file.hip [file.hip ...]: a copy of libbmp_hip.so with the given sources recompiled
# under extra flags -> tools/lib_NAME.so (for A/B runs: BMP_LIB_PATH=$PWD/tools/lib_NAME.so python bench.py ...)
set -e
NAME=$1; FLAGS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/gcn-bmp_amd/csrc
TMP=$(mktemp -d)
OBJS=""
for s in $CSRC/*.hip; do
  b=$(basename $s)
  o=$CSRC/build/$b.o
  for v in "$@"; do
    if [ "$v" = "$b" ]; then
      o=$TMP/$b.o
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $FLAGS -I $CSRC -c $s -o $o 2>/dev/null
    fi
  done
  OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/lib_$NAME.so $OBJS
rm -rf $TMP
echo built tools/lib_$NAME.so
