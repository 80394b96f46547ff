#!/bin/bash
# Per-row latency of the streaming kernel B with few waves per SIMD: one 1080p frame as 138 strips x N bands
out=gpurun_out/lone; mkdir -p $out
for sr in 1080 540 270 135 90 45; do
  for lib in "$@"; do
    QCNN_SEG_ROWS=$sr timeout -k 10 100 python3 - $lib <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from qcnn_gpu_amd import capi, synth
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "..", sys.argv[1], "libqcnn.so")
from qcnn_gpu_amd.qvrcnn import qvrcnn
h, w = 1080, 1920
net = qvrcnn(0, 1, 1, h, w)
net.load_static_para(synth.synth_model(37).to_nchw_vect_c_bytes())
net.load_data(synth.synth_frames(h, w, 37)[1][0])
for _ in range(20): net.forward_blu()
net.synchronize(); net.set_timing(True); net.kernel_times_us(reset=True)
for _ in range(50): net.forward_blu()
ka, kb, n = net.kernel_times_us()
sr = int(os.environ["QCNN_SEG_ROWS"]); bands = (h + sr - 1) // sr
print("%-18s seg_rows %4d: %5d waves (%.2f per SIMD), k_c3c4 %7.1f us, %6.1f ns per row of a wave" % (sys.argv[1], sr, 138 * bands, 138 * bands / 1024, kb, kb * 1000 / (sr + 5)))
PY
  done
done
