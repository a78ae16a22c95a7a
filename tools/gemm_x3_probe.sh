#!/bin/bash
# Developer: ablations of the f16x3 K loop on the GPU box (run from the repo root; variant libs are built here with hipcc).
#   gpurun --timeout 900 -- 'bash tools/gemm_x3_probe.sh'
mkdir -p gpurun_out/x3v
python3 tools/gemm_x3_power.py 2>&1 | tee gpurun_out/x3_power_product.log
for v in NOFRAG NODMA NOBAR NOMFMA "NOFRAG -DZH_X3_NODMA" "NOFRAG -DZH_X3_NODMA -DZH_X3_NOBAR"; do
  name=$(echo "$v" | sed 's/ -DZH_X3_/_/g')
  bash tools/build_variant_lib.sh gpurun_out/x3v/lib_$name.so "-DZH_X3_$v" gemm_x3.hip > /dev/null 2>&1
  ZUTIS_HIP_LIB=$PWD/gpurun_out/x3v/lib_$name.so python3 tools/gemm_x3_power.py 2>&1 | tee gpurun_out/x3_power_$name.log
done
rm -rf gpurun_out/x3v
