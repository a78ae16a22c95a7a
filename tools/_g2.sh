cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python3 tools/gemm_small_bench.py all > gpurun_out/r04/gemm_small_bench.txt 2>&1
tail -40 gpurun_out/r04/gemm_small_bench.txt
