#!/bin/bash
# PMC passes for the C5 (irls_batch_kernel) and C4 (ms_fit_kernel) kernels: instruction counts, busy cycles and HBM
# traffic, each counter group in its own run (rocprofv3 --kernel-trace --pmc only, program directly after `--`).
#   bash scripts/profile_c4c5_pmc.sh r02      -> gpurun_out/<tag>_c4c5/ ; summary json printed and written there
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${TAG}_c4c5
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 170 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p${i}_c5" -o run -- python3 "$ROOT/scripts/dev_time_c5.py" > "$OUT/p${i}_c5.log" 2>&1
  echo "group $i ($grp) c5 rc=$?"
  # one capture per batch size: the kernel name does not say how many points a dispatch fitted
  for sz in 8192 65536 262144; do
    timeout -k 10 170 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p${i}_mstart_$sz" -o run -- python3 "$ROOT/scripts/dev_time_mstart.py" $sz > "$OUT/p${i}_mstart_$sz.log" 2>&1
    echo "group $i ($grp) mstart $sz rc=$?"
  done
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, statistics, sys
out, tag = sys.argv[1], sys.argv[2]
res = {}
import re
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    m = re.search(r"p\d+_mstart_(\d+)", f)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void gslnls::", "")
        if not (k.startswith("irls_batch_kernel") or k.startswith("ms_fit_kernel")):
            continue
        if m:
            k += " @ %s points per dispatch" % m.group(1)
        res.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
summ = {k: {c: {"median_per_dispatch": statistics.median(v), "dispatches": len(v)} for c, v in d.items()} for k, d in res.items()}
for k, d in summ.items():
    if "FETCH_SIZE" in d:
        # gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM section); KB units
        d["hbm_bytes_per_dispatch_corrected"] = 2 * 1024 * d["FETCH_SIZE"]["median_per_dispatch"] + 1024 * d.get("WRITE_SIZE", {"median_per_dispatch": 0})["median_per_dispatch"]
json.dump(summ, open(os.path.join(out, "%s_c4c5_pmc.json" % tag), "w"), indent=1, sort_keys=True)
print(json.dumps(summ, indent=1, sort_keys=True))
PY
