#!/bin/bash
# Round 5: one image (config 3) with the one-round split-pair GEMM tiles as selected until round 5 (variant library, built by
#   tools/build_variant_lib.sh tools/_abl/libzh_r4tiles.so "-DZH_X3_ROUND4_SMALL_TILES" gemm_x3.hip
# ) against this build, ABAB on one box.  usage (on the GPU box): bash tools/c3_small_tiles_ab.sh
for i in 1 2; do
  echo "== round-4 small tiles"; ZUTIS_HIP_LIB=$PWD/tools/_abl/libzh_r4tiles.so python tools/c3_bench.py 2>&1 | grep "480x640\|427x640"
  echo "== this build"; python tools/c3_bench.py 2>&1 | grep "480x640\|427x640"
done
