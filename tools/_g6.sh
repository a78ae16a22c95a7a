cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
ZUTIS_HIP_LIB=$PWD/tools/_abl/libzutis_probe.so timeout 600 python3 tools/gemm_x3_stamp.py b1 0 1288 965 3064 > gpurun_out/r04/stamp_b1.txt 2>&1
timeout 600 python3 tools/gemm_small_bench.py warm > gpurun_out/r04/gemm_warm.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04/tr_graph -- python3 tools/c3_trace_run.py graph 8 > gpurun_out/r04/tr_graph.log 2>&1
python3 tools/trace_list.py gpurun_out/r04/tr_graph im2col 1 > gpurun_out/r04/c3_graph_list2.txt 2>&1
rm -rf gpurun_out/r04/tr_graph
timeout 2400 python3 -m pytest tests/ -q -m gpu > gpurun_out/r04/pytest_gpu_2.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/stamp_b1.txt | cut -c1-220; grep -v amdgpu gpurun_out/r04/gemm_warm.txt | cut -c1-200; tail -3 gpurun_out/r04/c3_graph_list2.txt; tail -15 gpurun_out/r04/pytest_gpu_2.txt
