#!/usr/bin/env python3
"""Diagnostic: absolute timeline of kernel A's twelve waves from a -DQCNN_STAMPS build (STAMPS_LIB names the build
directory under qcnn_gpu_amd/): the stamps of iterations 6..8 of every wave of workgroup 0, relative to the first
stamp of producer 0's iteration 6.  Producer stamps: loop top, tile produced, end; consumer stamps: loop top, first
reads issued / deferred store done, one per kernel column of the matrix phase, end."""
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
PROD, CONS = 3, int(os.environ.get("STAMPS_CONS", 8))
t0 = int(buf[0][6 * PROD])
for wv in range(12):
    per = PROD if wv < 4 else CONS
    line = []
    for step in (6, 7, 8, 9):
        b = step * PROD if wv < 4 else 2 + (step - 1) * CONS
        line.append(" ".join(f"{int(buf[wv][b + i]) - t0:6d}" for i in range(per)))
    print(f"wave {wv:2d}: " + " | ".join(line))
