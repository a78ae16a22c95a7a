cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
echo product > gpurun_out/r04/abl.txt; timeout 300 python3 tools/gemm_small_bench.py abl >> gpurun_out/r04/abl.txt 2>&1
for v in NOMFMA NOMFMA_NOFRAG NODMA NOMFMA_NOFRAG_NOBAR; do echo $v >> gpurun_out/r04/abl.txt; ZUTIS_HIP_LIB=$PWD/tools/_abl/lib_$v.so timeout 300 python3 tools/gemm_small_bench.py abl >> gpurun_out/r04/abl.txt 2>&1; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/tr_attn -- python3 tools/attn_small_bench.py > gpurun_out/r04/attn_small_bench2.txt 2>&1
cp $(find gpurun_out/r04/tr_attn -name "*kernel_stats.csv" | head -1) gpurun_out/r04/attn_small_kernel_stats.csv; rm -rf gpurun_out/r04/tr_attn
timeout 1200 python3 -m pytest tests/test_precision_gpu.py -x -q -m gpu -k "few_row or sum_layernorm or split_k_planes or every_tile_variant or x3_matches_float64" > gpurun_out/r04/pytest_new.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/abl.txt | cut -c1-250; cat gpurun_out/r04/attn_small_bench2.txt | tail -8; cut -c1-120 gpurun_out/r04/attn_small_kernel_stats.csv | head; tail -5 gpurun_out/r04/pytest_new.txt
