#!/usr/bin/env python3
"""Diagnostic: runs the -DQCNN_STAMPS build (make -C qcnn_gpu_amd/csrc stamps) on a 1080p frame and
prints, per wave of workgroup 0, the cycles between consecutive in-kernel stamps of kernel A."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qcnn_gpu_amd import capi, synth  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "qcnn_gpu_amd", os.environ.get("STAMPS_LIB", "lib_stamps"), "libqcnn.so")
from qcnn_gpu_amd.qvrcnn import qvrcnn  # noqa: E402

h, w, qp = 1080, 1920, 37
net = qvrcnn(0, 1, 1, h, w)
net.load_static_para(synth.synth_model(qp).to_nchw_vect_c_bytes())
net.load_data(synth.synth_frames(h, w, qp)[1][0])
lib = capi.load()
buf = np.zeros((24, 128), np.uint64)
net.set_timing(True)
for it in range(20):
    net.forward_blu()
    net.synchronize()
    lib.qcnn_debug_stamps(net._h, buf.ctypes.data_as(C.c_void_p))
print("kernel times (us) k_c1c2, k_c3c4, passes:", net.kernel_times_us())
# rows 0..7: waves of kernel A (k_c1c2); rows 8..: waves of kernel B (k_c3c4)
t0 = buf[buf > 0].min()
for wv in range(24):
    row = buf[wv]
    n = int((row > 0).sum())
    if n == 0:
        continue
    rel = (row[:n] - t0).astype(np.int64)
    d = np.diff(rel)
    print(f"wave {wv:2d}: {n} stamps, start {rel[0]}, end {rel[-1]}")
    print("   deltas:", " ".join(str(int(x)) for x in d[:40]))
