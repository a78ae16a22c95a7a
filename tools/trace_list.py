"""Developer tool: the launches of a rocprofv3 kernel trace between two marker kernels, in order, with duration, gap to the
previous launch and grid size.  usage: trace_list.py <dir> <first-kernel-substring> [occurrence-from-the-end=1] [count=400]"""
import csv, glob, sys
root, first = sys.argv[1], sys.argv[2]
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cnt = int(sys.argv[4]) if len(sys.argv) > 4 else 400
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
i0 = idx[-occ]
i1 = idx[-occ + 1] if occ > 1 else len(rows)
sel = rows[i0:min(i1, i0 + cnt)]
prev = None
busy = 0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    prev = e
    busy += e - s
    wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "?"
    gx = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
    try:
        nblk = int(gx) // int(wg)
    except Exception:
        nblk = -1
    print("%8.2f us  gap %7.2f  blocks %6d x %4s  lds %6s  %s" % ((e - s) / 1e3, gap, nblk, wg, r.get("LDS_Block_Size", "?"), r["Kernel_Name"][:110]))
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
if occ > 1 and i1 < len(rows) and i1 == i0 + len(sel):
    print("idle until the next %s: %.2f us" % (first, (int(rows[i1]["Start_Timestamp"]) - prev) / 1e3))
print("launches %d  busy %.3f ms  span %.3f ms" % (len(sel), busy / 1e6, span / 1e6))
