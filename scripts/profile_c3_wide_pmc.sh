#!/bin/bash
# PMC passes for the C3 kernels (glm_pass_kernel EVAL / JTJU, glm_jtj_mfma_kernel) and the wide dense pass
# (wide_pass_kernel): HBM traffic (FETCH_SIZE / WRITE_SIZE in separate runs) and VALU / MFMA instruction counts.
#   bash scripts/profile_c3_wide_pmc.sh r03     -> gpurun_out/<tag>_c3wide/ ; summary json printed and written there
# rocprofv3 --kernel-trace --pmc only, the program directly after `--`.
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${TAG}_c3wide
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  for wl in c3 wide; do
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p${i}_$wl" -o run -- python3 "$ROOT/scripts/dev_time_$wl.py" > "$OUT/p${i}_$wl.log" 2>&1
    echo "group $i ($grp) $wl rc=$?"
  done
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, statistics, sys
out, tag = sys.argv[1], sys.argv[2]
res = {}
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void gslnls::", "")
        if not any(t in k for t in ("glm_", "wide_")):
            continue
        # the GLM pass kernel runs in two modes with the same name: keep them apart by grid size is not possible, so by
        # dispatch order within the run -- dev_time_c3.py times EVAL first, then JTJU; the summary keeps all values
        res.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
summ = {k: {c: {"median_per_dispatch": statistics.median(v), "min": min(v), "max": max(v), "dispatches": len(v)}
            for c, v in d.items()} for k, d in res.items()}
for k, d in summ.items():
    if "FETCH_SIZE" in d:
        # gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM section); KB units
        d["hbm_bytes_per_dispatch_corrected"] = 2 * 1024 * d["FETCH_SIZE"]["median_per_dispatch"] + 1024 * d.get("WRITE_SIZE", {"median_per_dispatch": 0})["median_per_dispatch"]
json.dump(summ, open(os.path.join(out, "%s_c3_wide_pmc.json" % tag), "w"), indent=1, sort_keys=True)
print(json.dumps(summ, indent=1, sort_keys=True))
PY
