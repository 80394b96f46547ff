#!/usr/bin/env python3
"""Diagnostic: runs the -DQCNN_STAMPS build (make -C qcnn_gpu_amd/csrc stamps) on a batch of 1080p frames and
prints the cycles between consecutive in-kernel stamps of the four waves of one workgroup of the streaming
kernel B (per step: head; per row: first MFMA, requant done, row end)."""
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
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 16
net = qvrcnn(0, 1, 1, h, w)
net.load_static_para(synth.synth_model(qp).to_nchw_vect_c_bytes())
lib = capi.load()
frame = synth.synth_frames(h, w, qp)[1][0]
d_in, d_out = C.c_void_p(), C.c_void_p()
capi.check(lib.qcnn_dev_alloc(C.byref(d_in), nf * h * w))
capi.check(lib.qcnn_dev_alloc(C.byref(d_out), nf * h * w))
for i in range(nf):
    capi.check(lib.qcnn_dev_h2d(C.c_void_p(d_in.value + i * h * w), frame.ctypes.data, h * w))
buf = np.zeros((24, 128), np.uint64)
net.set_timing(True)
for it in range(5):
    net.forward_blu_batch(d_in.value, d_out.value, nf, h * w)
    net.synchronize()
    lib.qcnn_debug_stamps(net._h, buf.ctypes.data_as(C.c_void_p))
print("kernel times (us) k_c1c2, k_c3c4, passes:", net.kernel_times_us())
for wv in range(8, 12):
    row = buf[wv]
    n = int((row > 0).sum())
    if n == 0:
        continue
    rel = (row[:n] - row[0]).astype(np.int64)
    d = np.diff(rel)
    print(f"wave {wv - 8}: {n} stamps, total {rel[-1]}")
    # 1 + 3 * R stamps per step
    per = 1 + 3 * 6
    for st in range(0, n - 1, per):
        print("   step:", " ".join(str(int(x)) for x in d[st:st + per]))
