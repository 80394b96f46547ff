#!/bin/bash
# Runs on the GPU box: kernel-trace statistics of the default bench command, then counter passes of a
# short bench run (each --pmc set in a run of its own).  Results land under gpurun_out/prof/.
#   tools/profile_r02.sh [quick]      quick: skip the kernel-trace statistics of the full bench
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SHORT="--no-cpu-baseline --batch 64 --prewarm 0.5 --steps 2 --warmup 1"
if [ "$1" != "quick" ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
  echo "stats pass done"
fi
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $SHORT > "$OUT/bench_sq.json" 2> "$OUT/sq.err"
echo "sq pass done"
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d "$OUT/pmc_lds" -- python3 "$ROOT/bench.py" $SHORT > "$OUT/bench_lds.json" 2> "$OUT/lds.err"
echo "lds pass done"
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LEVEL_WAVES --kernel-trace --output-format csv -d "$OUT/pmc_act" -- python3 "$ROOT/bench.py" $SHORT > "$OUT/bench_act.json" 2> "$OUT/act.err" || echo "act pass failed"
echo "act pass done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $SHORT > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $SHORT > "$OUT/bench_write.json" 2> "$OUT/write.err"
echo "traffic passes done"
find "$OUT" -name "*.csv" | head -30
