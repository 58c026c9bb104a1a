#!/bin/bash
# developer aid: the kernel timeline of the matrix path's trial steps at p = 99 (rocprofv3 --kernel-trace; start / end of every
# kernel of the last fit, microseconds from the first): where a 184 us trial step goes, launch by launch
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/trace_matrix_step
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o run -- python3 "$ROOT/scripts/dev_time_matrix_path.py" ${1:-33} ${2:-3000} > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the fused fits come before the stepwise ones: take the window of the 4th fit = the last fused one -- simplest: the first
# 400 kernels after the 3rd occurrence of cholb_init following a bd_publish gap ... print a slice instead
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gslnls::", "") for r in rows]
# find the start of the last "fused" fit: fused fits use bd_trial_kernel; take the last 60 kernels before the last bd_trial_kernel
idx = [i for i, n in enumerate(names) if n.startswith("bd_trial")]
last = idx[-1]
lo = max(0, last - 40)
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
for i in range(lo, min(len(rows), last + 8)):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    print("%8.1f us  +gap %5.1f  dur %6.1f  %s  grid %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, names[i][:60], rows[i].get("Grid_Size", "")))
    prev_end = e
PY
