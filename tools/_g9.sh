cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 600 python3 tools/c3_bench.py > gpurun_out/r04/c3_bench_3.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04/tr_graph -- python3 tools/c3_trace_run.py graph 8 > gpurun_out/r04/tr_graph.log 2>&1
python3 tools/trace_list.py gpurun_out/r04/tr_graph im2col 1 > gpurun_out/r04/c3_graph_list3.txt 2>&1
rm -rf gpurun_out/r04/tr_graph
timeout 900 python3 bench.py > gpurun_out/r04/bench_default_1.json 2> gpurun_out/r04/bench_default_1.err
timeout 2400 python3 -m pytest tests/ -q -m gpu > gpurun_out/r04/pytest_gpu_4.txt 2>&1
tail -4 gpurun_out/r04/c3_bench_3.txt; tail -2 gpurun_out/r04/c3_graph_list3.txt; tail -c 1500 gpurun_out/r04/bench_default_1.json; tail -3 gpurun_out/r04/bench_default_1.err; tail -8 gpurun_out/r04/pytest_gpu_4.txt
