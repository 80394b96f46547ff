#!/bin/bash
# Runs on the GPU box: kernel-trace statistics of the default bench command, counter passes of a short bench run
# (each --pmc set in a run of its own), kernel-trace statistics of the dynamic-quantisation and float32 passes, and a
# power / clock trace of the sustained loop.  Results land under gpurun_out/prof/ (and gpurun_out/prof/extra/).
#   tools/profile_r03.sh [quick]      quick: skip the kernel-trace statistics of the full bench
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof"
rm -rf "$OUT"; mkdir -p "$OUT/extra"
cd /tmp && export TMPDIR=/tmp
SHORT="--no-cpu-baseline --no-e2e --batch 64 --prewarm 0.5 --steps 2 --warmup 1"
if [ "$1" != "quick" ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-e2e > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
  echo "stats pass done"
fi
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $SHORT > "$OUT/bench_sq.json" 2> "$OUT/sq.err"
echo "sq pass done"
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d "$OUT/pmc_lds" -- python3 "$ROOT/bench.py" $SHORT > "$OUT/bench_lds.json" 2> "$OUT/lds.err"
echo "lds pass done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $SHORT > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $SHORT > "$OUT/bench_write.json" 2> "$OUT/write.err"
echo "traffic passes done"
# the dynamic-quantisation pass and the float32 configuration on this build
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/dyn" -- python3 "$ROOT/tools/bench_dynamic.py" --only dynamic,dynamic_layered,static_layered > "$OUT/extra/bench_dynamic.json" 2> "$OUT/dyn.err"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/float" -- python3 "$ROOT/tools/bench_dynamic.py" --only float32 > "$OUT/extra/bench_float.json" 2> "$OUT/float.err"
echo "dynamic / float passes done"
cd "$ROOT"
AB_FRAMES=256 AB_PREWARM=3 AB_STEPS=40 timeout -k 10 120 python3 tools/power_trace.py "$OUT/extra/power_shipped" -- python3 tools/ab_bench.py --inline lib > "$OUT/extra/power_shipped.out" 2>&1
echo "power trace done"
find "$OUT" -name "*_kernel_stats.csv" | head
