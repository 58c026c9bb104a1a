#!/bin/bash
# Round profile on the MI355X box: the bench line, rocprofv3 kernel stats of the same command, and the two
# PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass).  Usage (through gpurun):
#   bash scripts/profile_round.sh r01b
# Writes under gpurun_out/<tag>/ ; scripts/summarise_profile.py turns that into the files kept in profiles/.
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 "$ROOT/bench.py" > "$OUT/bench_line.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$ROOT/bench.py" --steps 100 --warmup 10 --headline-only --no-cpu-baseline > "$OUT/stats.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o run -- python3 "$ROOT/bench.py" --steps 20 --warmup 2 --headline-only --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o run -- python3 "$ROOT/bench.py" --steps 20 --warmup 2 --headline-only --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1
# the whole line incl. the C3 / C4 / C5 side measurements, under the tracer
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_full" -o run -- python3 "$ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/stats_full.log" 2>&1
echo "profiles done"
python3 "$ROOT/scripts/summarise_profile.py" "$OUT" "$TAG"
cp "$(find "$OUT/stats_full" -name '*kernel_stats.csv' | head -1)" "$OUT/summary/${TAG}_full_bench_kernel_stats.csv" 2>/dev/null
