cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
./tools/_abl/dma_stream > gpurun_out/r04/dma_stream.txt 2>&1
timeout 300 python3 tools/_diag1.py > gpurun_out/r04/diag1.txt 2>&1
timeout 600 python3 tools/c3_bench.py > gpurun_out/r04/c3_bench_2.txt 2>&1
timeout 2400 python3 -m pytest tests/ -q -m gpu > gpurun_out/r04/pytest_gpu_3.txt 2>&1
cat gpurun_out/r04/dma_stream.txt; grep -v amdgpu gpurun_out/r04/diag1.txt | tail -8; tail -4 gpurun_out/r04/c3_bench_2.txt; tail -8 gpurun_out/r04/pytest_gpu_3.txt
