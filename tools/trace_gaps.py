"""Developer tool: per-kernel durations and inter-kernel gaps of the LAST n launches in a rocprofv3 kernel trace."""
import csv, glob, sys, collections
root, n = sys.argv[1], int(sys.argv[2])
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(dur(r) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = [int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) for i in range(len(rows) - 1)]
print("kernels busy %.3f ms, span %.3f ms, avg gap %.2f us, avg dur %.2f us" % (tot / 1e6, span / 1e6, sum(gaps) / len(gaps) / 1e3, tot / len(rows) / 1e3))
d = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = r["Kernel_Name"][:60]; d[k][0] += 1; d[k][1] += dur(r)
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%-60s %4d  %7.1f us avg  %8.1f us total" % (k, v[0], v[1] / v[0] / 1e3, v[1] / 1e3))
