cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python3 tools/gemm_small_bench.py enc > gpurun_out/r04/gemm_small_bench3.txt 2>&1
timeout 1500 python3 -m pytest tests/test_precision_gpu.py -q -m gpu -k "every_tile_variant or x2_is_bitwise or few_row or split_k" > gpurun_out/r04/pytest_k64.txt 2>&1
timeout 600 python3 -m pytest tests/test_e2e_gpu.py -q -m gpu -k "batch_invariance" > gpurun_out/r04/pytest_inv.txt 2>&1
grep -v amdgpu gpurun_out/r04/gemm_small_bench3.txt | cut -c1-330; tail -6 gpurun_out/r04/pytest_k64.txt; tail -4 gpurun_out/r04/pytest_inv.txt
