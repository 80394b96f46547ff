import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import c_oracle
from qcnn_gpu_amd import capi, synth
from qcnn_gpu_amd.qvrcnn import qvrcnn
h, w, qp = int(sys.argv[1]), int(sys.argv[2]), 37
m = synth.synth_model(qp)
mb = m.to_nchw_vect_c_bytes()
y = synth.synth_frames(h, w, qp)[1][0]
rec_o, taps = c_oracle.forward_blu(mb, y, True)
with qvrcnn(0, 1, 1, h, w) as net:
    net.load_static_para(mb)
    net.load_data(y); net.forward_blu(); net.synchronize()
    rec = net.download()
print("mismatches vs oracle", int((rec != rec_o).sum()), "of", h * w)
conc2 = taps["conc2"]                                  # [12][h][w][4]
c2 = conc2.transpose(0, 3, 1, 2).reshape(48, h, w).astype(np.int64)
w4 = m.layers[5].w.astype(np.int64)[0]                 # [48][3][3]
b4 = int(m.layers[5].b[0]); mul, shift = m.layers[5].mul, m.layers[5].shift
pad = np.pad(c2, ((0, 0), (1, 1), (1, 1)))
def run(taps_used, transpose=False, chan_sel=None):
    acc = np.zeros((h, w), np.int64)
    for r, s in taps_used:
        ww = w4[:, s, r] if transpose else w4[:, r, s]
        if chan_sel is not None:
            ww = ww * chan_sel
        acc += np.tensordot(ww, pad[:, r:r + h, s:s + w], axes=(0, 0))
    t = ((acc + b4) * mul + (1 << (shift - 1))) >> shift
    return np.clip(y.astype(np.int64) + t, 0, 255).astype(np.uint8)
allt = [(r, s) for r in range(3) for s in range(3)]
print("check full == oracle:", int((run(allt) != rec_o).sum()))
for name, tp in [("all", allt), ("s=0 only", [(r, 0) for r in range(3)]), ("s<=1", [(r, s) for r in range(3) for s in range(2)]),
                 ("r=1 only", [(1, s) for s in range(3)]), ("none", [])]:
    print(name, "mismatch vs gpu:", int((run(tp) != rec).sum()))
print("transposed:", int((run(allt, True) != rec).sum()))
for t in range(3):
    sel = np.zeros(48, np.int64); sel[16 * t:16 * t + 16] = 1
    print("only channels of tile", t, ":", int((run(allt, False, sel) != rec).sum()), " without:", int((run(allt, False, 1 - sel) != rec).sum()))
np.save("gpurun_out/dbg_rec.npy", rec)

# ---- hypotheses on the C3 stage ----
conc1 = taps["conc1"].transpose(0, 3, 1, 2).reshape(48, h, w).astype(np.int64)
def requant(u, lp):
    bias = (1 << (lp.shift - 1)) // lp.mul
    q = ((u + bias) * lp.mul) >> lp.shift
    return np.where(u > lp.blu, 127, np.where(u < 0, 0, q))
def conv(x, lp, rmap=lambda r: r, smap=lambda s: s, rowshift=0):
    wgt = lp.w.astype(np.int64)                       # [K][C][ks][ks]
    K, C, ks, _ = wgt.shape
    p = ks // 2
    xp = np.pad(x, ((0, 0), (p + 2, p + 2), (p, p)))
    out = np.zeros((K, h, w), np.int64)
    for r in range(ks):
        for s in range(ks):
            out += np.tensordot(wgt[:, :, rmap(r), smap(s)], xp[:, 2 + r + rowshift:2 + r + rowshift + h, s:s + w], axes=(1, 0))
    return out + lp.b.astype(np.int64)[:, None, None]
def c4(c2v):
    padv = np.pad(c2v, ((0, 0), (1, 1), (1, 1)))
    acc = np.zeros((h, w), np.int64)
    for r in range(3):
        for s in range(3):
            acc += np.tensordot(w4[:, r, s], padv[:, r:r + h, s:s + w], axes=(0, 0))
    t = ((acc + b4) * mul + (1 << (shift - 1))) >> shift
    return np.clip(y.astype(np.int64) + t, 0, 255).astype(np.uint8)
L3, L4 = m.layers[3], m.layers[4]
def variant(**kw31):
    kw32 = kw31.pop("kw32", {})
    a = requant(conv(conc1, L3, **kw31), L3)
    b = requant(conv(conc1, L4, **kw32), L4)
    return c4(np.concatenate([a, b], 0))
print("C3 baseline == oracle:", int((variant() != rec_o).sum()))
flip = lambda r: 2 - r
for name, kw in [("rows flipped", dict(rmap=flip)), ("cols flipped", dict(smap=flip)), ("both flipped", dict(rmap=flip, smap=flip)),
                 ("c31 rowshift +1", dict(rowshift=1)), ("c31 rowshift -1", dict(rowshift=-1)),
                 ("c32 rowshift +1", dict(kw32=dict(rowshift=1))), ("c32 rowshift -1", dict(kw32=dict(rowshift=-1)))]:
    print(name, "mismatch vs gpu:", int((variant(**kw) != rec).sum()))

# ---- conc2 dump of a diagnostic build (make alt ALT=-DQCNN_DEBUG_CONC2) ----
if os.environ.get("DBG_CONC2"):
    import ctypes as C
    L = C.CDLL(os.path.join(os.path.dirname(capi.LIB_PATH), "..", "lib_alt", "libqcnn.so"))
    hnd = C.c_void_p()
    L.qcnn_create.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_void_p)]
    assert L.qcnn_create(0, 1, 1, h, w, C.byref(hnd)) == 0
    L.qcnn_load_static_mem.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    assert L.qcnn_load_static_mem(hnd, mb, len(mb)) == 0
    L.qcnn_upload.argtypes = [C.c_void_p, C.c_void_p]
    L.qcnn_forward_blu.argtypes = [C.c_void_p]
    L.qcnn_debug_conc2.argtypes = [C.c_void_p, C.c_void_p]
    assert L.qcnn_upload(hnd, y.ctypes.data) == 0 and L.qcnn_forward_blu(hnd) == 0
    dump = np.empty((h, w, 48), np.int8)
    assert L.qcnn_debug_conc2(hnd, dump.ctypes.data) == 0
    # internal position 12g + 4t + i  <->  channel 16t + 4g + i
    pos = np.arange(48); g_, t_, i_ = pos // 12, (pos % 12) // 4, pos % 4
    ch = 16 * t_ + 4 * g_ + i_
    got = np.empty((48, h, w), np.int64)
    got[ch] = dump.transpose(2, 0, 1)
    bad = np.argwhere(got != c2)
    print("conc2 mismatches:", len(bad), "of", got.size)
    if len(bad):
        print(" by channel tile:", np.bincount(bad[:, 0] // 16, minlength=3).tolist())
        print(" by row:", np.bincount(bad[:, 1], minlength=h).tolist())
        print(" by col mod 14:", np.bincount(bad[:, 2] % 14, minlength=14).tolist())
        print(" first:", bad[:8].tolist(), [(int(got[tuple(b)]), int(c2[tuple(b)])) for b in bad[:8]])
