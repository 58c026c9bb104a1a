#!/bin/bash
# developer aid: the wide path's trial step (p = 32; n = 1e5, 1e6) under the contraction loop's developer switches and the
# row emitter's group size -- one process per variant (the switches are read once), one line per variant
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
run() {
  local name=$1; shift
  local out
  out=$(env "$@" timeout -k 10 300 python3 scripts/dev_time_wide.py 2>/dev/null | python3 -c '
import json,sys
d=json.load(sys.stdin)
print(" ".join("%s: %.1f us/step (%d steps)" % (k, 1e3*v["ms_per_trial_step"], v["steps"]) for k,v in d.items() if k.startswith("n=")))')
  echo "$name | $out"
}
if [ $# -gt 0 ]; then run "$@"; exit 0; fi
run "default" X=1
run "unroll 8" GSLNLS_RTC_EXTRA_FLAGS="-DGSLNLS_WIDE_CHUNK_UNROLL=8"
run "unroll 16" GSLNLS_RTC_EXTRA_FLAGS="-DGSLNLS_WIDE_CHUNK_UNROLL=16"
run "unroll 2" GSLNLS_RTC_EXTRA_FLAGS="-DGSLNLS_WIDE_CHUNK_UNROLL=2"
run "prio 3" GSLNLS_RTC_EXTRA_FLAGS="-DGSLNLS_WIDE_MFMA_PRIO=3"
run "prio 3 unroll 16" GSLNLS_RTC_EXTRA_FLAGS="-DGSLNLS_WIDE_MFMA_PRIO=3 -DGSLNLS_WIDE_CHUNK_UNROLL=16"
run "group 3" GSLNLS_RTC_SINK_GROUP=3
run "group 4" GSLNLS_RTC_SINK_GROUP=4
run "group 10" GSLNLS_RTC_SINK_GROUP=10
run "group 16" GSLNLS_RTC_SINK_GROUP=16
