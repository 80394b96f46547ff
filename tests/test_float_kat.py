"""Known answers of the float32-configuration oracle (oracle/qcnn_oracle_float.c), derived by hand from the
source formulas of FLOAT_CONFIG (inference/cnn.cu:114-127, 470-502, inference/mat.cu:160-176), plus a
cross-check of its convolution against torch's float32 conv2d.  No GPU.

PARITY UNPINNED: the reference holds no float model and no output of this configuration.
"""
import ctypes as C

import numpy as np

from oracle import c_oracle
from qcnn_gpu_amd import synth


def test_model_size():
    # (1600+64 + 18432+32 + 25600+16 + 6912+16 + 1536+32 + 432+1) floats
    assert c_oracle.lib().orf_model_bytes() == 4 * 54673 == 218692
    assert len(synth.synth_float_model(37)) == 218692


def test_ppro_known_answers():
    x = np.array([0, 128, 255, 1], np.uint8)
    out = np.empty(4, np.float32)
    c_oracle.lib().orf_ppro(c_oracle._p(x), c_oracle._p(out), 4)
    want = np.array([np.float32(-128) / np.float32(255), 0, np.float32(127) / np.float32(255),
                     np.float32(-127) / np.float32(255)], np.float32)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))


def test_apply_res_known_answers():
    one = c_oracle.lib().orf_apply_res_one
    assert one(100, 0.0) == 100                   # 100 + 0.5 truncates to 100
    assert one(100, 1 / 255) == 101
    assert one(100, 0.49 / 255) == 100            # 100.99 truncates
    assert one(100, 0.51 / 255) == 101
    assert one(100, -0.51 / 255) == 99            # 99.99 -> 99
    assert one(10, -10.7 / 255) == 0              # -0.2 truncates toward zero -> 0
    assert one(0, -1.0) == 0 and one(250, 0.1) == 255 and one(255, 1000.0) == 255 and one(0, -1000.0) == 0


def test_conv_identity_and_zero_padding():
    h, w = 5, 7
    x = np.arange(h * w, dtype=np.float32).reshape(1, h, w)
    # centre tap 1: u = x + b
    wt = np.zeros((2, 1, 3, 3), np.float32)
    wt[:, 0, 1, 1] = 1
    b = np.array([0.5, -100], np.float32)
    u = c_oracle.conv_float(x, wt, b, relu=False)
    assert np.array_equal(u[0], x[0] + 0.5) and np.array_equal(u[1], x[0] - 100)
    assert np.array_equal(c_oracle.conv_float(x, wt, b, relu=True)[1], np.zeros((h, w), np.float32))
    # all-ones 3x3 on ones counts the taps inside the frame: 4 in a corner, 6 on an edge, 9 inside
    ones = np.ones((1, h, w), np.float32)
    u = c_oracle.conv_float(ones, np.ones((1, 1, 3, 3), np.float32), np.zeros(1, np.float32), relu=False)[0]
    assert u[0, 0] == 4 and u[0, 3] == 6 and u[2, 0] == 6 and u[2, 3] == 9 and u[4, 6] == 4
    # cross-correlation, not convolution (inference/cnn.cu:48): the tap right of centre picks x[y][x+1]
    wt = np.zeros((1, 1, 3, 3), np.float32)
    wt[0, 0, 1, 2] = 1
    u = c_oracle.conv_float(x, wt, np.zeros(1, np.float32), relu=False)[0]
    assert np.array_equal(u[:, :-1], x[0][:, 1:]) and np.all(u[:, -1] == 0)


def test_conv_against_torch():
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(5)
    for cin, cout, ks in ((1, 64, 5), (64, 32, 3), (64, 16, 5), (48, 32, 1), (48, 1, 3)):
        x = rng.standard_normal((cin, 9, 13)).astype(np.float32)
        w = (rng.standard_normal((cout, cin, ks, ks)) * 0.1).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        u = c_oracle.conv_float(x, w, b, relu=True)
        t = F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), torch.from_numpy(b), padding=(ks - 1) // 2).relu()[0].numpy()
        s = c_oracle.conv_float(np.abs(x), np.abs(w), np.abs(b), relu=False)
        assert np.all(np.abs(u - t) <= 4e-6 * s), (cin, cout, ks)
        u32 = c_oracle.conv_float(x, w, b, relu=True, acc32=True)
        assert np.all(np.abs(u - u32) <= 4e-6 * s)


def test_hwcn2nchw():
    h, w, c, n = 3, 3, 2, 4
    src = np.arange(h * w * c * n, dtype=np.float32)
    out = np.empty(n * c * h * w, np.float32)
    assert c_oracle.lib().orf_hwcn2nchw(c_oracle._p(src), h, w, c, n, c_oracle._p(out)) == out.size
    assert np.array_equal(out.reshape(n, c, h, w), src.reshape(h, w, c, n).transpose(3, 2, 0, 1))


def test_forward_is_the_layer_chain():
    """orf_forward against the same chain spelled with conv_float (inference/qvrcnn.cu:244-285)."""
    mb = synth.synth_float_model(32)
    _, anc = synth.synth_frames(20, 24, 32, 1)
    y = anc[0]
    rec, t = c_oracle.forward_float(mb, y, True)
    f = np.frombuffer(mb, np.float32)
    ws, bs, off = [], [], 0
    for ks, cin, cout in zip(c_oracle.F_KS, c_oracle.F_CIN, c_oracle.COUT):
        n = ks * ks * cin * cout
        ws.append(f[off:off + n].reshape(cout, cin, ks, ks))
        off += n
        bs.append(f[off:off + cout])
        off += cout
    xp = ((y.astype(np.float32) - np.float32(128)) / np.float32(255))[None]
    u1 = c_oracle.conv_float(xp, ws[0], bs[0], True)
    c1 = np.concatenate([c_oracle.conv_float(u1, ws[1], bs[1], True), c_oracle.conv_float(u1, ws[2], bs[2], True)])
    c2 = np.concatenate([c_oracle.conv_float(c1, ws[3], bs[3], True), c_oracle.conv_float(c1, ws[4], bs[4], True)])
    u4 = c_oracle.conv_float(c2, ws[5], bs[5], False)[0]
    assert np.array_equal(t["x_ppro"], xp[0]) and np.array_equal(t["u1"], u1)
    assert np.array_equal(t["conc1"], c1) and np.array_equal(t["conc2"], c2) and np.array_equal(t["u4"], u4)
    one = c_oracle.lib().orf_apply_res_one
    want = np.array([[one(int(y[i, j]), float(u4[i, j])) for j in range(y.shape[1])] for i in range(y.shape[0])], np.uint8)
    assert np.array_equal(rec, want)
    assert np.abs(rec.astype(int) - y.astype(int)).max() > 0   # the network does something
