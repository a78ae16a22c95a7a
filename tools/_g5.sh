cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
echo product > gpurun_out/r04/abl.txt; timeout 300 python3 tools/gemm_small_bench.py abl >> gpurun_out/r04/abl.txt 2>&1
for v in NOMFMA NOMFMA_NOFRAG NODMA NOMFMA_NOFRAG_NOBAR; do echo $v >> gpurun_out/r04/abl.txt; ZUTIS_HIP_LIB=$PWD/tools/_abl/lib_$v.so timeout 300 python3 tools/gemm_small_bench.py abl >> gpurun_out/r04/abl.txt 2>&1; done
timeout 300 python3 tools/attn_small_bench.py > gpurun_out/r04/attn_small_bench3.txt 2>&1
timeout 600 python3 tools/c3_bench.py > gpurun_out/r04/c3_bench_1.txt 2>&1
timeout 2400 python3 -m pytest tests/ -x -q -m gpu > gpurun_out/r04/pytest_gpu_1.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/abl.txt | cut -c1-250; cat gpurun_out/r04/attn_small_bench3.txt | tail -8; tail -5 gpurun_out/r04/c3_bench_1.txt; tail -15 gpurun_out/r04/pytest_gpu_1.txt
