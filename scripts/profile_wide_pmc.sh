#!/bin/bash
# PMC passes for the wide path's one-launch-per-trial-step kernel (wide_step_kernel<ModelJit, 0, 32> at n = 1e5 and 1e6,
# scripts/dev_time_wide.py): HBM traffic (FETCH_SIZE / WRITE_SIZE in separate runs), VALU and MFMA instruction counts, waves.
#   bash scripts/profile_wide_pmc.sh r05     -> gpurun_out/<tag>_wide/ ; summary json printed and written there
# rocprofv3 --kernel-trace --pmc only, the program directly after `--`.
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${TAG}_wide
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p${i}" -o run -- python3 "$ROOT/scripts/dev_time_wide.py" > "$OUT/p${i}.log" 2>&1
  echo "group $i ($grp) rc=$?"
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, statistics, sys
out, tag = sys.argv[1], sys.argv[2]
res = {}
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void gslnls::", "")
        if "wide_" not in k:
            continue
        # two problem sizes share the kernel's name: keep them apart by grid size
        key = "%s grid=%s" % (k, row.get("Grid_Size", "?"))
        res.setdefault(key, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
summ = {k: {c: {"median_per_dispatch": statistics.median(v), "min": min(v), "max": max(v), "dispatches": len(v)}
            for c, v in d.items()} for k, d in res.items()}
for k, d in summ.items():
    if "FETCH_SIZE" in d:
        # gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM section); KB units
        d["hbm_bytes_per_dispatch_corrected"] = 2 * 1024 * d["FETCH_SIZE"]["median_per_dispatch"] + 1024 * d.get("WRITE_SIZE", {"median_per_dispatch": 0})["median_per_dispatch"]
json.dump(summ, open(os.path.join(out, "%s_wide_pmc.json" % tag), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in summ.items() if "step" in k}, indent=1, sort_keys=True)[:6000])
PY
