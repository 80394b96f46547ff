#!/usr/bin/env python3
"""Diagnostic: how long each of kernel A's workgroups runs in a launch of 16 frames (-DQCNN_STAMPS build; STAMPS_LIB
names the build directory), per role -- wave 0 (producer), 4 (older consumer), 8 (younger consumer) -- with the tile
count of the workgroup."""
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
net.set_timing(True)
for it in range(8):
    net.forward_blu_batch(d_in.value, d_out.value, nf, h * w)
    net.synchronize()
print("kernel times (us) k_c1c2, k_c3c4, passes:", net.kernel_times_us())
blk = np.zeros((16384, 3), np.uint64)
lib.qcnn_debug_blocks(net._h, blk.ctypes.data_as(C.c_void_p))
for role, name in ((0, "producer 0"), (1, "older consumer 4"), (2, "younger consumer 8")):
    r = blk[8192 + role * 2048: 8192 + role * 2048 + 256]
    ok = r[:, 1] > 0
    d = (r[ok, 1] - r[ok, 0]).astype(np.int64)
    n = r[ok, 2].astype(np.int64)
    q = np.percentile(d, [0, 5, 50, 95, 100])
    print("%s: %d workgroups, tiles %d..%d, duration ticks min %d p5 %d median %d p95 %d max %d" % ((name, ok.sum(), n.min(), n.max()) + tuple(q)))
    if role == 0:
        order = np.argsort(d)
        print("   slowest workgroups:", [(int(i), int(d[i])) for i in order[-6:]], " fastest:", [(int(i), int(d[i])) for i in order[:4]])
        per_tile = d / np.maximum(n, 1)
        print("   ticks per tile: min %d median %d max %d" % (per_tile.min(), np.median(per_tile), per_tile.max()))

# cost of a tile by kind, from the workgroups' durations: least squares over the workgroups of
#   duration = c_int * interior + c_edge * (tiles on the left / right frame edge) + c_tb * (top / bottom tiles of inner columns)
TH, TW = 20, 32
tiles_x, tiles_y = (w + TW - 1) // TW, (h + TH - 1) // TH
tpf = tiles_x * tiles_y
ntiles = tpf * nf
chunk = (ntiles + 255) // 256
r = blk[8192: 8192 + 256]
rows, ys = [], []
for b in range(256):
    k0 = b * chunk
    n = max(0, min(chunk, ntiles - k0))
    if n == 0 or r[b, 1] == 0:
        continue
    kinds = [0, 0, 0]
    for k in range(k0, k0 + n):
        t = k % tpf
        tx, ty = t // tiles_y, t % tiles_y
        if tx == 0 or tx == tiles_x - 1:
            kinds[1] += 1
        elif ty == 0 or ty == tiles_y - 1:
            kinds[2] += 1
        else:
            kinds[0] += 1
    rows.append(kinds + [1])
    ys.append(int(r[b, 1] - r[b, 0]))
A = np.array(rows, float)
sol, *_ = np.linalg.lstsq(A, np.array(ys, float), rcond=None)
print("ticks per tile: interior %.0f, left/right edge %.0f, top/bottom %.0f, per workgroup %.0f" % tuple(sol))
pred = A @ sol
print("fit residual: rms %.0f ticks of %.0f" % (np.sqrt(np.mean((pred - ys) ** 2)), np.mean(ys)))
