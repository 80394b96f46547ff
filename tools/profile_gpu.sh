#!/bin/bash
# Runs on the GPU box: kernel-trace statistics of the default bench command, then two counter passes
# (HBM-side read and write bytes) of a short bench run.  Results land under gpurun_out/prof/.
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 2 --warmup 1 > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 2 --warmup 1 > "$OUT/bench_write.json" 2> "$OUT/write.err"
find "$OUT" -name "*.csv" | head -20
