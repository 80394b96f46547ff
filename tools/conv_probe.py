#!/usr/bin/env python3
"""Diagnostic: times the layer-level convolution (qcnn_op_conv_static) of every VRCNN layer shape on a
1080p image, for one or more builds of the library (directories under qcnn_gpu_amd/, default "lib").

    python tools/conv_probe.py lib lib_x1 lib_x2
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
from qcnn_gpu_amd import capi
capi.LIB_PATH = os.path.join(%(root)r, "qcnn_gpu_amd", %(lib)r, "libqcnn.so")
lib = capi.load()
h, w = 1080, 1920
def dev(n):
    p = C.c_void_p(); capi.check(lib.qcnn_dev_alloc(C.byref(p), n)); capi.check(lib.qcnn_dev_memset(p, 1, n)); return p
out = []
for name, ks, cin, cout in (("C1", 5, 1, 64), ("C2_1", 3, 64, 32), ("C2_2", 5, 64, 16), ("C3_1", 3, 48, 16), ("C3_2", 1, 48, 32), ("C4", 3, 48, 1)):
    cp = (cin + 3) // 4 * 4
    x, wt, b, u = dev(cp * h * w), dev(ks * ks * cp * cout), dev(4 * cout), dev(4 * cout * h * w)
    capi.check(lib.qcnn_dev_memset(b, 0, 4 * cout))
    for _ in range(3): capi.check(lib.qcnn_op_conv_static(x, wt, b, u, h, w, cin, cout, ks, None))
    capi.check(lib.qcnn_dev_sync()); t0 = time.perf_counter()
    for _ in range(20): capi.check(lib.qcnn_op_conv_static(x, wt, b, u, h, w, cin, cout, ks, None))
    capi.check(lib.qcnn_dev_sync()); dt = (time.perf_counter() - t0) / 20
    out.append("%%s %%6.1f us" %% (name, dt * 1e6))
    for p in (x, wt, b, u): capi.check(lib.qcnn_dev_free(p))
print("%%-12s " %% %(lib)r + "  ".join(out))
"""
for name in sys.argv[1:] or ["lib"]:
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "lib": name}], capture_output=True, text=True, timeout=300)
    sys.stdout.write(r.stdout if r.returncode == 0 else f"{name}: FAILED\n{r.stderr[-2000:]}\n")
    sys.stdout.flush()
