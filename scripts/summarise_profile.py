"""Condense one scripts/profile_round.sh run into the small files kept under profiles/:
<tag>_bench_line.json, <tag>_bench_kernel_stats.csv, <tag>_pmc_traffic.json"""
import csv
import glob
import json
import os
import statistics
import sys

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(out, "summary")
os.makedirs(dst, exist_ok=True)


def find(pattern):
    hits = glob.glob(os.path.join(out, pattern), recursive=True)
    return hits[0] if hits else None


line = open(os.path.join(out, "bench_line.json")).read().strip().splitlines()
if line:
    open(os.path.join(dst, "%s_bench_line.json" % tag), "w").write(line[-1] + "\n")
ks = find("stats/**/*kernel_stats.csv")
if ks:
    open(os.path.join(dst, "%s_bench_kernel_stats.csv" % tag), "w").write(open(ks).read())
res = {}
for name, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = find("%s/**/*counter_collection.csv" % d)
    if not f:
        continue
    per = {}
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != name:
            continue
        k = row["Kernel_Name"]
        short = k.split("(")[0].replace("void gslnls::", "")
        per.setdefault(short, []).append(float(row["Counter_Value"]))
    for k, v in per.items():
        res.setdefault(k, {})["%s_KB_median" % name] = statistics.median(v)
        res[k]["%s_KB_mean" % name] = sum(v) / len(v)
        res[k]["launches"] = len(v)
for k, v in res.items():
    if "FETCH_SIZE_KB_median" in v:
        # gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM section)
        v["bytes_per_launch_corrected"] = 2 * 1024 * v["FETCH_SIZE_KB_median"] + 1024 * v.get("WRITE_SIZE_KB_median", 0.0)
json.dump(res, open(os.path.join(dst, "%s_pmc_traffic.json" % tag), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in res.items() if "lm_step" in k}, indent=1))
