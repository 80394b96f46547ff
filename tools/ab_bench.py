#!/usr/bin/env python3
"""Diagnostic: times the two fused kernels of one build of the library on 16 resident 1080p frames.

    python tools/ab_bench.py lib lib_alt lib lib_alt      # directories under qcnn_gpu_amd/, run in turn

Each named build is loaded in a child process (one library per process) and reports the mean launch
time of k_c1c2 / k_c3c4 over 20 steps, so that two builds can be compared on the same device in the
same session (device-to-device differences are larger than most kernel changes).  Short bursts ride the boost
clock: AB_FRAMES=512 AB_PREWARM=2 gives the sustained figures of bench.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
from qcnn_gpu_amd import capi, synth
capi.LIB_PATH = os.path.join(%(root)r, "qcnn_gpu_amd", %(lib)r, "libqcnn.so")
from qcnn_gpu_amd.qvrcnn import qvrcnn
h, w, qp, B = (2160, 3840, 32, 5) if os.environ.get("AB_4K") else (1080, 1920, 37, 16)
B = int(os.environ.get("AB_FRAMES", B))              # resident frames per step
steps = int(os.environ.get("AB_STEPS", 20))
prewarm = float(os.environ.get("AB_PREWARM", 0))     # seconds of untimed steps first (sustained clocks)
lib = capi.load()
net = qvrcnn(0, 1, 1, h, w)
if os.environ.get("AB_EXACT"): net.set_exact(True)
net.load_static_para(synth.synth_model(qp).to_nchw_vect_c_bytes())
anc = synth.synth_frames(h, w, qp, frames=4)[1]
d_in, d_out = C.c_void_p(), C.c_void_p()
capi.check(lib.qcnn_dev_alloc(C.byref(d_in), B * h * w)); capi.check(lib.qcnn_dev_alloc(C.byref(d_out), B * h * w))
for i in range(B):
    f = np.ascontiguousarray(anc[i %% 4]); capi.check(lib.qcnn_dev_h2d(C.c_void_p(d_in.value + i * h * w), f.ctypes.data, h * w))
for _ in range(5): net.forward_blu_batch(d_in.value, d_out.value, B, h * w)
t0 = time.perf_counter()
while time.perf_counter() - t0 < prewarm:
    net.forward_blu_batch(d_in.value, d_out.value, B, h * w); capi.check(lib.qcnn_dev_sync())
capi.check(lib.qcnn_dev_sync()); net.set_timing(True); net.kernel_times_us(reset=True)
t0 = time.perf_counter()
for _ in range(steps): net.forward_blu_batch(d_in.value, d_out.value, B, h * w)
capi.check(lib.qcnn_dev_sync()); dt = time.perf_counter() - t0
ka, kb, n = net.kernel_times_us()
ka, kb = ka * n / steps, kb * n / steps  # per step (a step of more than one launch group)
import zlib
out = np.empty(B * h * w, np.uint8); capi.check(lib.qcnn_dev_d2h(out.ctypes.data, d_out, B * h * w))
print("%%-10s k_c1c2 %%7.1f us/frame  k_c3c4 %%7.1f us/frame  total %%8.1f Mpx/s  crc32(out) %%08x" %% (%(lib)r, ka / B, kb / B, steps * B * h * w / dt / 1e6, zlib.crc32(out.tobytes())))
"""

if len(sys.argv) == 3 and sys.argv[1] == "--inline":  # this process itself (for use under rocprofv3)
    exec(CHILD % {"root": ROOT, "lib": sys.argv[2]})
    sys.exit(0)
for name in sys.argv[1:] or ["lib"]:
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "lib": name}], capture_output=True, text=True, timeout=300)
    sys.stdout.write(r.stdout if r.returncode == 0 else f"{name}: FAILED\n{r.stderr[-2000:]}\n")
    sys.stdout.flush()
