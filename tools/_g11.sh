cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 2400 python3 -m pytest tests/ -q -m gpu > gpurun_out/r04/pytest_gpu_6.txt 2>&1
timeout 600 python3 tools/c3_bench.py > gpurun_out/r04/c3_bench_5.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04/tr_graph -- python3 tools/c3_trace_run.py graph 8 > gpurun_out/r04/tr_graph.log 2>&1
python3 tools/trace_list.py gpurun_out/r04/tr_graph im2col 1 > gpurun_out/r04/c3_graph_list4.txt 2>&1
rm -rf gpurun_out/r04/tr_graph
tail -8 gpurun_out/r04/pytest_gpu_6.txt; tail -4 gpurun_out/r04/c3_bench_5.txt; tail -22 gpurun_out/r04/c3_graph_list4.txt | cut -c1-150
