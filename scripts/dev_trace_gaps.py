"""print per-kernel durations and gaps from a rocprofv3 --kernel-trace csv"""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
print(len(rows), "kernels")
prev_end = None
import collections
dur = collections.defaultdict(list); gaps = []
for i, (s, e, k) in enumerate(rows):
    dur[k].append(e - s)
    if prev_end is not None:
        gaps.append(s - prev_end)
    prev_end = e
for k, v in dur.items():
    v = sorted(v)
    print("%-60s n=%d min=%.2fus med=%.2fus max=%.2fus" % (k, len(v), v[0] / 1e3, v[len(v) // 2] / 1e3, v[-1] / 1e3))
g = sorted(gaps)
print("gaps: med=%.2fus p10=%.2f p90=%.2f" % (g[len(g) // 2] / 1e3, g[len(g) // 10] / 1e3, g[9 * len(g) // 10] / 1e3))
# print a window of one solve
start = max(0, len(rows) - 40)
for i in range(start, len(rows)):
    s, e, k = rows[i]
    print("%3d dur=%7.2fus gap_before=%8.2fus %s" % (i, (e - s) / 1e3, (s - rows[i - 1][1]) / 1e3 if i else 0, k[:40]))
