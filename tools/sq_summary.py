"""Summarise rocprofv3 --pmc passes of SQ counters for ONE kernel into a JSON (profiles/r0N_attention_pmc*.json).

    python tools/sq_summary.py <dir with *counter_collection.csv> <kernel-name substring> <out.json> "<command>" [duration_us]
"""
import csv, glob, json, sys, collections
root, key, out, cmd = sys.argv[1:5]
dur = float(sys.argv[5]) if len(sys.argv) > 5 else None
acc, n = collections.defaultdict(float), collections.defaultdict(set)
name = None
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            name = r["Kernel_Name"]
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
c = {k: round(v / max(1, len(n[k]))) for k, v in sorted(acc.items())}
d = {}
if c.get("SQ_INSTS_MFMA"):
    d["valu_insts_per_mfma"] = round((c.get("SQ_INSTS_VALU", 0) - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"], 2)
    d["valu_note"] = "SQ_INSTS_VALU counts MFMAs too on gfx950: (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA"
if c.get("SQ_WAVE_CYCLES"):
    w = c["SQ_WAVE_CYCLES"]
    d["wave_cycle_split"] = {k2: round(c.get(k1, 0) / w, 3) for k1, k2 in (("SQ_ACTIVE_INST_ANY", "active_inst"), ("SQ_WAIT_INST_ANY", "wait_inst_any(issue stall)"),
                                                                          ("SQ_WAIT_ANY", "wait_any(waitcnt/barrier)"))}
if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("SQ_BUSY_CYCLES"):
    # SQ_BUSY_CYCLES is summed over the shader engines' SQs; per-SIMD MFMA utilisation = MFMA busy cycles / (1024 SIMDs x kernel cycles)
    d["mfma_busy_cycles_per_launch"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]
    if dur:
        d["mfma_pipe_busy_fraction"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * dur * 1e-6 * 2.0e9), 3)
        d["mfma_pipe_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x ~2.0 GHz), duration = %.1f us per launch (kernel trace of the same run)" % dur
json.dump({"command": cmd, "kernel": name, "counters_per_launch": c, "derived": d}, open(out, "w"), indent=1)
print(json.dumps(d))
