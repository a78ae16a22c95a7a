#!/bin/bash
# Round 5: the headline step at `fast` (three launch plans in flight, then one stream) and one image of config 3 at `fast`, with the plain-fp16
# one-round tiles as selected until round 5 (variant library, built by
#   tools/build_variant_lib.sh tools/_abl/libzh_r4tiles_f16.so "-DZH_F16_ROUND4_SMALL_TILES" gemm.hip
# ) against this build, ABAB on one box.  usage (on the GPU box): bash tools/fast_small_tiles_ab.sh
Q="--precision fast --no-second-precision --no-live-traffic --no-io-rates --no-batch1 --no-configs --no-cpu-baseline --no-torch-gpu-baseline --steps 40"
val() { python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('timed_outputs_checked'))"; }
for i in 1 2; do
  echo "== round-4 tiles, 3 in flight: $(ZUTIS_HIP_LIB=$PWD/tools/_abl/libzh_r4tiles_f16.so python bench.py $Q 2>/dev/null | val)"
  echo "== this build,    3 in flight: $(python bench.py $Q 2>/dev/null | val)"
  echo "== round-4 tiles, one stream:  $(ZUTIS_HIP_LIB=$PWD/tools/_abl/libzh_r4tiles_f16.so python bench.py $Q --inflight 1 2>/dev/null | val)"
  echo "== this build,    one stream:  $(python bench.py $Q --inflight 1 2>/dev/null | val)"
done
echo "== config 3, round-4 tiles"; ZUTIS_HIP_LIB=$PWD/tools/_abl/libzh_r4tiles_f16.so python tools/c3_bench.py 2>&1 | grep "fast"
echo "== config 3, this build"; python tools/c3_bench.py 2>&1 | grep "fast"
