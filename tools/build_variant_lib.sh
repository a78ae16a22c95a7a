#!/bin/bash
# Developer A/B: an alternate libzutis_hip built with extra -D flags on selected sources, for same-box comparisons through
# ZUTIS_HIP_LIB (box-to-box variance on this pool is +-3 %).   usage: tools/build_variant_lib.sh OUT.so "-DFLAG ..." file.hip [file.hip ...]
set -e
OUT=$1; FLAGS=$2; shift 2
mkdir -p "$(dirname "$OUT")"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
SKIP=""
for f in "$@"; do
  b=$(basename "$f" .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $FLAGS -c "$ROOT/zutis_amd/csrc/$b.hip" -o "$TMP/$b.o"
  SKIP="$SKIP $b.o"
done
OBJS=""
for o in "$ROOT"/zutis_amd/_obj/*.o; do
  case " $SKIP gemm_rs.o " in *" $(basename "$o") "*) ;; *) OBJS="$OBJS $o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS "$TMP"/*.o
rm -rf "$TMP"
echo "built $OUT"
