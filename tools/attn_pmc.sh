#!/bin/bash
# SQ counter passes of the attention kernel (run on the GPU box from the repo root): two --pmc passes per (library, shape), the
# kernel trace of the same run for the duration, summarised by tools/sq_summary.py into gpurun_out/<tag>_attention_pmc_<...>.json.
#   usage: tools/attn_pmc.sh <tag> <shape: enc|cross> <x3|f16> [lib.so]
TAG=$1; SHAPE=$2; MODE=$3; LIB=$4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
[ -n "$LIB" ] && export ZUTIS_HIP_LIB=$LIB
D=gpurun_out/pmc_$TAG
rm -rf $D; mkdir -p $D
ARG=""; [ "$MODE" = "x3" ] && ARG="x3"
rocprofv3 --kernel-trace --stats --output-format csv -d $D/t -- python3 tools/attn_pmc_run.py $SHAPE $ARG > /dev/null 2>&1
DUR=$(python3 - <<PY
import csv, glob
for f in glob.glob("$D/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_f16_kernel" in r["Name"]:
            print(float(r["AverageNs"]) / 1e3); break
PY
)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $D/p1 -- python3 tools/attn_pmc_run.py $SHAPE $ARG > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $D/p2 -- python3 tools/attn_pmc_run.py $SHAPE $ARG > /dev/null 2>&1
python3 tools/sq_summary.py $D attn_f16_kernel gpurun_out/${TAG}.json "rocprofv3 --pmc <8 SQ counters> --kernel-trace -- python3 tools/attn_pmc_run.py $SHAPE $ARG (two passes; lib=${LIB:-product}; per launch, mean of 10)" $DUR
rm -rf $D
