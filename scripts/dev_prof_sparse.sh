#!/bin/bash
# per-kernel times of the sparse large path (run through gpurun)
cd /tmp && export TMPDIR=/tmp
for args in "2000000 200000 16" "4000000 100000 4"; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/sp
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sp -o run -- python3 $GRAFT_REPO_ROOT/scripts/dev_time_sparse.py $args 2>/dev/null | tail -1
  python3 - <<PY
import csv,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/sp/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("   ", r["Name"][:60], r["Calls"], r["AverageNs"])
PY
done
