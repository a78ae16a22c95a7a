cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 2400 python3 -m pytest tests/ -q -m gpu > gpurun_out/r04/pytest_gpu_5.txt 2>&1
timeout 600 python3 tools/c3_bench.py > gpurun_out/r04/c3_bench_4.txt 2>&1
timeout 900 python3 bench.py --no-cpu-baseline --no-torch-gpu-baseline --no-live-traffic --no-io-rates > gpurun_out/r04/bench_default_2.json 2> gpurun_out/r04/bench_default_2.err
tail -12 gpurun_out/r04/pytest_gpu_5.txt; tail -4 gpurun_out/r04/c3_bench_4.txt; tail -c 600 gpurun_out/r04/bench_default_2.json
