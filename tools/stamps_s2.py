#!/usr/bin/env python3
"""Diagnostic: runs a -DQCNN_STAMPS build (make -C qcnn_gpu_amd/csrc variant NAME=lib_stamps ALT=-DQCNN_STAMPS;
STAMPS_LIB names the build directory) on a batch of 1080p frames and prints
  * the ticks one wave of kernel B (two-tile form, QCNN_STAMP_BLOCK picks the workgroup) spends per phase of a row,
    averaged over its untested rows -- 0 top-of-row wait, 1 C4 MFMAs + LDS reads + third kernel row, 2 gathers +
    second kernel row, 3 first kernel row, 4 C3_2 + end of the requantiser, 5 stores, 6 row requests, 7 between rows
    (not with -DQCNN_STAMPS_NOPH, which leaves the rows alone and stamps only the start and end of every workgroup);
  * when and where every workgroup of the last launch ran: residency per compute unit, the distribution of the
    workgroups' durations, by frame, strip, band, XCD and SIMD, and whether the two waves of a SIMD take equally long."""
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
for it in range(8):
    net.forward_blu_batch(d_in.value, d_out.value, nf, h * w)
    net.synchronize()
    lib.qcnn_debug_stamps(net._h, buf.ctypes.data_as(C.c_void_p))
print("kernel times (us) k_c1c2, k_c3c4, passes:", net.kernel_times_us())
ph = buf[20, :8].astype(np.float64)
rows = float(buf[20, 8])
if rows:
    print("rows %d, cycles per row by phase:" % rows, " ".join("%.0f" % (x / rows) for x in ph), " total %.0f" % (ph.sum() / rows))

# residency of kernel B's workgroups (single waves): per compute unit, how many ran at the same time
blk = np.zeros((16384, 3), np.uint64)
lib.qcnn_debug_blocks(net._h, blk.ctypes.data_as(C.c_void_p))
blk = blk[blk[:, 1] > 0]
t0, t1, hw = blk[:, 0].astype(np.int64), blk[:, 1].astype(np.int64), blk[:, 2]
hwid, xcc = (hw & 0xffffffff).astype(np.int64), ((hw >> 32) & 0xf).astype(np.int64)
cu, sh, se, simd = (hwid >> 8) & 0xf, (hwid >> 12) & 1, (hwid >> 13) & 0x7, (hwid >> 4) & 3
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("workgroups %d, distinct compute units %d, distinct (unit, SIMD) %d" % (len(blk), len(set(key)), len(set(key * 4 + simd))))
base = t0.min()
print("start ticks: min 0, median %d, max %d; end ticks: min %d, median %d, max %d; duration median %d" % (
    np.median(t0 - base), (t0 - base).max(), (t1 - base).min(), np.median(t1 - base), (t1 - base).max(), np.median(t1 - t0)))
per = {}
for k, a_, b_ in zip(key, t0, t1):
    per.setdefault(int(k), []).append((a_, b_))
counts = sorted(len(v) for v in per.values())
print("workgroups per compute unit: min %d median %d max %d" % (counts[0], counts[len(counts) // 2], counts[-1]))
# peak concurrency per unit
peaks = []
for k, v in per.items():
    ev = sorted([(a_, 1) for a_, _ in v] + [(b_, -1) for _, b_ in v])
    c = m = 0
    for _, d in ev:
        c += d
        m = max(m, c)
    peaks.append(m)
peaks.sort()
print("peak resident workgroups per compute unit: min %d median %d max %d" % (peaks[0], peaks[len(peaks) // 2], peaks[-1]))

# duration by workgroup: distribution, and which units are the slow ones (unit = (b & 7) * per_xcd + (b >> 3))
dur = (t1 - t0)
q = np.percentile(dur, [0, 5, 25, 50, 75, 95, 100])
print("duration ticks: min %d p5 %d p25 %d median %d p75 %d p95 %d max %d" % tuple(q))
allblk = np.zeros((16384, 3), np.uint64)
lib.qcnn_debug_blocks(net._h, allblk.ctypes.data_as(C.c_void_p))
valid = np.nonzero(allblk[:, 1] > 0)[0]
grid = len(valid)
per_xcd = (grid + 7) // 8
nstrips = (w + 29) // 30
d_all = (allblk[valid, 1] - allblk[valid, 0]).astype(np.int64)
unit = (valid & 7) * per_xcd + (valid >> 3)
strip = unit % nstrips
rest = unit // nstrips
nseg = grid // (nstrips * nf) if grid >= nstrips * nf else 1
seg = rest % max(nseg, 1)
for name, sel in (("edge strips", (strip == 0) | (strip == nstrips - 1)), ("inner strips", (strip > 0) & (strip < nstrips - 1))):
    if sel.any():
        print("%s: %d workgroups, duration median %d max %d" % (name, sel.sum(), np.median(d_all[sel]), d_all[sel].max()))
for sg in range(max(nseg, 1)):
    sel = seg == sg
    if sel.any():
        print("band %d: duration median %d max %d" % (sg, np.median(d_all[sel]), d_all[sel].max()))
# per XCD: spread of start and end
xc = ((allblk[valid, 2] >> 32) & 0xf).astype(np.int64)
for x in sorted(set(xc)):
    sel = xc == x
    a0 = allblk[valid, 0][sel].astype(np.int64)
    a1 = allblk[valid, 1][sel].astype(np.int64)
    b0 = a0.min()
    print("XCD %d: %d workgroups, starts within %d ticks, ends from %d to %d" % (x, sel.sum(), a0.max() - b0, a1.min() - b0, a1.max() - b0))

hw_all = (allblk[valid, 2] & 0xffffffff).astype(np.int64)
simd_all, cu_all, se_all = (hw_all >> 4) & 3, (hw_all >> 8) & 0xf, (hw_all >> 13) & 7
frame_all = rest // max(nseg, 1)
def by(name, keyv):
    out = []
    for kx in sorted(set(keyv.tolist())):
        sel = keyv == kx
        out.append("%d:%d" % (kx, np.median(d_all[sel]) / 1000))
    print("median duration (k ticks) by %s: %s" % (name, " ".join(out)))
by("XCD", xc)
by("SIMD", simd_all)
by("SE", se_all)
by("CU", cu_all)
by("frame", frame_all)
by("strip", strip)
fast = d_all < 500000
print("fast workgroups: %d of %d" % (fast.sum(), len(d_all)))
# are the two waves of a SIMD alike?
pair = {}
for kx, dd in zip((xc * 8 + se_all) * 64 + cu_all * 4 + simd_all, d_all):
    pair.setdefault(int(kx), []).append(int(dd))
same = sum(1 for v in pair.values() if len(v) == 2 and (v[0] < 500000) == (v[1] < 500000))
print("SIMDs with two workgroups: %d, of them both fast or both slow: %d" % (sum(1 for v in pair.values() if len(v) == 2), same))
