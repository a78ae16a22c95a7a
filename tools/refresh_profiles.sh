#!/bin/bash
# Regenerates the rocprofv3 summaries and bench lines under profiles/ (run on the GPU box from the repo root, e.g.
#   gpurun --timeout 3000 -- 'bash tools/refresh_profiles.sh r06'
# then copy gpurun_out/<round>_* into profiles/).  One rocprofv3 --kernel-trace --stats pass per configuration; the PMC traffic
# passes are the ones bench.py spawns itself (roofline.traffic of the default line).  The headline precision is `exact` (bench.py's
# default); `fast` is profiled as the second precision.
R=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
mkdir -p gpurun_out
Q="--no-cpu-baseline --no-torch-gpu-baseline --no-live-traffic --no-second-precision --no-io-rates --no-batch1 --no-configs --steps 10 --warmup 3"
prof() {
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_$name -- python3 bench.py "$@" > gpurun_out/p_$name.log 2>&1
  f=$(find gpurun_out/p_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${R}_${name}_kernel_stats.csv
  rm -rf gpurun_out/p_$name
}
prof bench_exact $Q --inflight 1
prof bench_exact_inflight $Q
prof bench_fast $Q --inflight 1 --precision fast
prof c4 $Q --workload c4
prof c5 $Q --workload c5 --inflight 1
# batch 1 (config 3's shape through the drop-in module): per-kernel stats and the ordered launch list of ONE hipGraph-replayed forward + predict
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_c3 -- python3 tools/c3_trace_run.py graph 12 > gpurun_out/p_c3.log 2>&1
f=$(find gpurun_out/p_c3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${R}_c3_graph_kernel_stats.csv
python3 tools/trace_list.py gpurun_out/p_c3 im2col 3 > gpurun_out/${R}_c3_launch_list.txt 2>&1
rm -rf gpurun_out/p_c3
python3 tools/b1_joint.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_b1_joint.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_bs -- python3 tools/bilateral_prof_run.py 1 > gpurun_out/p_bs.log 2>&1
f=$(find gpurun_out/p_bs -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${R}_bilateral_b1_kernel_stats.csv
rm -rf gpurun_out/p_bs
# the pseudo-label path (SelfMask T = 5505 + solver + resize), batch 4, default precision: per-kernel stats, and the SQ counters of its attention launch
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_sm -- python3 tools/selfmask_prof_run.py 4 exact > gpurun_out/p_sm.log 2>&1
f=$(find gpurun_out/p_sm -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${R}_selfmask_b4_exact_kernel_stats.csv
rm -rf gpurun_out/p_sm
bash tools/attn_pmc.sh ${R}_attention_pmc_selfmask_x3 selfmask4 x3
python3 bench.py 2> gpurun_out/${R}_bench.err | tail -1 > gpurun_out/${R}_bench.json
python3 bench.py --workload c4 2>> gpurun_out/${R}_bench.err | tail -1 > gpurun_out/${R}_bench_c4.json
python3 bench.py --workload c5 2>> gpurun_out/${R}_bench.err | tail -1 > gpurun_out/${R}_bench_c5.json
python3 bench.py --workload c3 2>> gpurun_out/${R}_bench.err | tail -1 > gpurun_out/${R}_bench_c3.json
python3 bench.py --inflight 1 --no-cpu-baseline --no-torch-gpu-baseline --no-live-traffic --no-second-precision --no-io-rates --no-batch1 --no-configs 2>> gpurun_out/${R}_bench.err | tail -1 > gpurun_out/${R}_bench_inflight1.json
head -c 600 gpurun_out/${R}_bench.json; echo
for f in gpurun_out/${R}_bench_c3.json gpurun_out/${R}_bench_c4.json gpurun_out/${R}_bench_c5.json gpurun_out/${R}_bench_inflight1.json; do python3 -c "
import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['unit'], d['ms_per_step'], d.get('roofline',{}) and d['roofline'].get('achieved'))"; done
