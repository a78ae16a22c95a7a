cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python3 tools/gemm_small_bench.py all > gpurun_out/r04/gemm_small_bench2.txt 2>&1
timeout 300 python3 tools/attn_small_bench.py > gpurun_out/r04/attn_small_bench.txt 2>&1
# counters of the batch-1 QKV / fc GEMM (auto tile): L2 hit rate, fetch bytes
for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r04/pmc_$n -- python3 tools/gemm_small_bench.py pmc > gpurun_out/r04/pmc_$n.log 2>&1
  python3 - <<PY > gpurun_out/r04/pmc_$n.txt 2>&1
import csv, glob, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r04/pmc_$n/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            d[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print(k, {c: (sum(x) / len(x), len(x)) for c, x in v.items()})
PY
  rm -rf gpurun_out/r04/pmc_$n
done
tail -30 gpurun_out/r04/gemm_small_bench2.txt; cat gpurun_out/r04/attn_small_bench.txt; cat gpurun_out/r04/pmc_*.txt
