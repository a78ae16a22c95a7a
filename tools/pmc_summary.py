"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/rNN_pmc_traffic.json.

    python tools/pmc_summary.py <dir with *counter_collection.csv> <out.json> "<command that was profiled>"

gfx950: FETCH_SIZE under-counts wide coalesced reads 2x (MI355X_MICROARCH.md, HBM / rocprofv3 section) -> HBM read bytes =
2 * FETCH_SIZE; WRITE_SIZE is exact; both are reported in KiB by the tool.  All GEMM template variants are also pooled
into one entry (`gemm_f16_kernel_all_variants`) = the dominant kernel of bench.py's roofline."""
import csv, glob, json, sys, collections

root, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k, c = r["Kernel_Name"], r["Counter_Name"]
        acc[k][c] += float(r["Counter_Value"])
        disp[k][c].add((f, r["Dispatch_Id"]))
kern = {}
pool = collections.defaultdict(float)
pool_n = collections.defaultdict(int)
for k in acc:
    e = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        if c in acc[k]:
            n = len(disp[k][c])
            e["dispatches"] = n
            e[c + "_MB_per_dispatch"] = round(acc[k][c] / n / 1024.0, 3)
            if "gemm_f16_kernel" in k:
                pool[c] += acc[k][c]
                pool_n[c] += n
    kern[k[:60]] = e
g = {}
if pool_n["FETCH_SIZE"] and pool_n["WRITE_SIZE"]:
    fetch = pool["FETCH_SIZE"] / pool_n["FETCH_SIZE"] * 1024.0
    write = pool["WRITE_SIZE"] / pool_n["WRITE_SIZE"] * 1024.0
    g = {"dispatches": pool_n["FETCH_SIZE"], "fetch_bytes_per_launch_raw": round(fetch), "write_bytes_per_launch": round(write),
         "hbm_bytes_per_launch": round(2 * fetch + write)}
json.dump({"command": cmd, "note": "gfx950: HBM read = 2*FETCH_SIZE (tool under-counts wide coalesced reads 2x), WRITE_SIZE exact; KiB in the tool output",
           "gemm_f16_kernel_all_variants": g, "kernels": kern}, open(out, "w"), indent=1)
print(json.dumps(g))
