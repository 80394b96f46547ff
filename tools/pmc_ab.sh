#!/bin/bash
# counters of kernel A for two builds
cd /tmp && export TMPDIR=/tmp
for lib in lib lib_v_l2st; do
  timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_ab_$lib -- python3 $GRAFT_REPO_ROOT/tools/ab_bench.py --inline $lib > $GRAFT_REPO_ROOT/gpurun_out/pmc_ab_$lib.txt 2>&1 || exit 1
done
