#!/bin/bash
# counters of the fused kernels for a list of builds (directories under qcnn_gpu_amd/), one rocprofv3 pass each:
#   tools/pmc_ab.sh "COUNTER ..." lib lib_v1 ...
cd /tmp && export TMPDIR=/tmp
CNT="$1"; shift
for lib in "$@"; do
  timeout -s KILL 120 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_ab_$lib -- python3 $GRAFT_REPO_ROOT/tools/ab_bench.py --inline $lib > $GRAFT_REPO_ROOT/gpurun_out/pmc_ab_$lib.txt 2>&1 || exit 1
  python3 $GRAFT_REPO_ROOT/tools/pmc_means.py $GRAFT_REPO_ROOT/gpurun_out/pmc_ab_$lib
done
