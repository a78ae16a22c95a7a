#!/bin/bash
# Round 6: upper bounds of the three GEMM leads of the round-5 review, measured with timing-only ablation libraries (garbage results)
# on ONE box, ABAB against the product library:
#   libzh_tailsplit.so   -DZH_ABL_TAIL_SPLIT        an ideal stream-K / fixed-split tail round (no reduction traffic)
#   libzh_skipstores.so  -DZH_ABL_SKIP_EPI_STORES   the epilogue's global stores gone (what wave-specialised store roles could hide at most)
#   libzh_skipepi.so     -DZH_ABL_SKIP_EPI          the whole epilogue gone (round 5's bound, for reference)
# Workloads: the headline step at `fast` (plain fp16 persistent tiles) with three plans in flight and on one stream; config 5 (ViT-L/14).
Q="--precision fast --steps 30 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-live-traffic --no-second-precision --no-io-rates --no-batch1 --no-configs"
val() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  for lib in product tailsplit skipstores ${LEAD_LIBS_EXTRA}; do
    if [ $lib = product ]; then unset ZUTIS_HIP_LIB; else export ZUTIS_HIP_LIB=$PWD/tools/_abl/libzh_$lib.so; fi
    echo "$lib: c2 fast 3 in flight $(python3 bench.py $Q 2>/dev/null | val) | one stream $(python3 bench.py $Q --inflight 1 2>/dev/null | val) | c5 fast (6 layers) $(python3 bench.py --workload c5 --c5-layers 6 --steps 4 --warmup 1 --no-cpu-baseline --no-second-precision 2>/dev/null | val)"
  done
done
unset ZUTIS_HIP_LIB
