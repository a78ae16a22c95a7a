cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python3 tools/c3_bench.py > gpurun_out/r04/c3_bench_base.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04/tr_graph -- python3 tools/c3_trace_run.py graph 8 > gpurun_out/r04/tr_graph.log 2>&1
python3 tools/trace_list.py gpurun_out/r04/tr_graph im2col 1 > gpurun_out/r04/c3_graph_list.txt 2>&1
python3 tools/trace_gaps.py gpurun_out/r04/tr_graph 2000 > gpurun_out/r04/c3_graph_gaps.txt 2>&1
rm -rf gpurun_out/r04/tr_graph
tail -3 gpurun_out/r04/c3_bench_base.txt; tail -3 gpurun_out/r04/c3_graph_list.txt
