#!/usr/bin/env python3
"""Diagnostic: timeline of kernel A's twelve waves from a -DQCNN_STAMPS build (STAMPS_LIB names the build
directory under qcnn_gpu_amd/).  Prints, for a few steady-state steps, every stamp of every wave of workgroup 0
relative to the start of the step (the earliest first stamp of the step over the waves)."""
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
PROD, CONS = 3, int(os.environ.get("STAMPS_CONS", 8))  # stamps per step of a producer / consumer wave
for step in (6, 7, 8):
    base = lambda wv: step * PROD if wv < 4 else 2 + (step - 1) * CONS  # (a consumer's step 0 has two stamps only)
    t0 = min(int(buf[wv][base(wv)]) for wv in range(12))
    print(f"step {step}")
    for wv in range(12):
        n = PROD if wv < 4 else CONS
        rel = [int(buf[wv][base(wv) + i]) - t0 for i in range(n + 1)]
        print(f"  wave {wv:2d}: " + " ".join(f"{x:6d}" for x in rel))
