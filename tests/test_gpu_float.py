"""GPU parity of the float32 configuration (qvrcnn::forward under FLOAT_CONFIG, inference/qvrcnn.cu:244-285)
through the C ABI against the CPU oracle.

Tolerance (float32 sums depend on the summation order, which the reference leaves to cuDNN; the oracle
accumulates in double): a convolution output may differ from the oracle by at most
    TOL * (sum |w|*|x| + |b|),   TOL = 4e-6 (about 64 float32 epsilons; the worst case of a 1600-term
                                  float32 chain is 1600 epsilons)
per element when both see the same input, and 4x that through the four-layer chain.  FLOAT_PPRO and applyRes
are element-wise and must be bit-exact on the same input.  Reconstructed pixels may differ by one grey level
where the residual sits on a rounding boundary: at most 0.1 % of the pixels, never more than 1.
PARITY UNPINNED by the reference (no float model, no outputs shipped).
"""
import ctypes as C

import numpy as np
import pytest

from oracle import c_oracle
from qcnn_gpu_amd import capi, synth
from qcnn_gpu_amd.qvrcnn import qvrcnn

pytestmark = pytest.mark.gpu
TOL = 4e-6
SHAPES = ((1, 64, 5), (64, 32, 3), (64, 16, 5), (48, 16, 3), (48, 32, 1), (48, 1, 3))


class Dev:
    """Device buffer through the C ABI's memory helpers."""

    def __init__(self, arr=None, nbytes=0):
        self.L = capi.load()
        self.p = C.c_void_p()
        self.n = arr.nbytes if arr is not None else nbytes
        capi.check(self.L.qcnn_dev_alloc(C.byref(self.p), max(self.n, 4)))
        if arr is not None:
            a = np.ascontiguousarray(arr)
            capi.check(self.L.qcnn_dev_h2d(self.p, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def get(self, dtype, shape):
        out = np.empty(shape, dtype)
        capi.check(self.L.qcnn_dev_sync())
        capi.check(self.L.qcnn_dev_d2h(out.ctypes.data_as(C.c_void_p), self.p, out.nbytes))
        return out

    def free(self):
        self.L.qcnn_dev_free(self.p)


def _conv_gpu(x, w, b, relu):
    L = capi.load()
    cout, cin, ks, _ = w.shape
    _, h, wd = x.shape
    dx, dw, db = Dev(x), Dev(w), Dev(b)
    du = Dev(nbytes=4 * cout * h * wd)
    capi.check(L.qcnn_dev_memset(du.p, 0xFF, du.n))
    capi.check(L.qcnn_op_conv_float(dx.p, dw.p, db.p, du.p, h, wd, cin, cout, ks, int(relu), None))
    u = du.get(np.float32, (cout, h, wd))
    for d in (dx, dw, db, du):
        d.free()
    return u


@pytest.mark.parametrize("cin,cout,ks", SHAPES)
@pytest.mark.parametrize("h,w", ((1, 1), (5, 7), (8, 32), (9, 33), (37, 53)))
def test_conv_float_layer_shapes(cin, cout, ks, h, w):
    rng = np.random.default_rng(1000 * cin + 10 * cout + ks + h)
    x = rng.standard_normal((cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, ks, ks)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    s = c_oracle.conv_float(np.abs(x), np.abs(wt), np.abs(b), relu=False)
    for relu in (False, True):
        u = _conv_gpu(x, wt, b, relu)
        o = c_oracle.conv_float(x, wt, b, relu)
        assert np.all(np.isfinite(u))
        bad = np.argwhere(np.abs(u - o) > TOL * s)
        assert bad.size == 0, f"{len(bad)} elements beyond tolerance, first {bad[:3].tolist()}"


def test_conv_float_exact_on_integers():
    """Small-integer operands make every partial sum exact in float32: the result must equal the oracle bit
    for bit whatever the order, which pins the k-index maps of the matrix instruction (asymmetric data)."""
    rng = np.random.default_rng(7)
    for cin, cout, ks in SHAPES + ((64, 64, 3), (48, 40, 5), (64, 7, 1)):
        x = rng.integers(-8, 9, (cin, 19, 45)).astype(np.float32)
        wt = rng.integers(-4, 5, (cout, cin, ks, ks)).astype(np.float32)
        b = rng.integers(-100, 100, cout).astype(np.float32)
        u = _conv_gpu(x, wt, b, False)
        o = c_oracle.conv_float(x, wt, b, False)
        assert np.array_equal(u, o), (cin, cout, ks, np.argwhere(u != o)[:3].tolist())


def test_conv_float_rejects_other_shapes():
    L = capi.load()
    d = Dev(nbytes=64)
    assert L.qcnn_op_conv_float(d.p, d.p, d.p, d.p, 4, 4, 3, 8, 3, 0, None) == -6
    assert L.qcnn_op_conv_float(d.p, d.p, d.p, d.p, 4, 4, 64, 65, 3, 0, None) == -6
    assert L.qcnn_op_conv_float(None, d.p, d.p, d.p, 4, 4, 64, 16, 3, 0, None) == -1
    d.free()


def test_ppro_and_apply_res_bit_exact():
    L = capi.load()
    rng = np.random.default_rng(3)
    n = 256 * 257 + 3
    x = rng.integers(0, 256, n).astype(np.uint8)
    x[:256] = np.arange(256)
    res = (rng.standard_normal(n) * 0.05).astype(np.float32)
    res[:8] = [0, 1 / 255, 0.49 / 255, 0.51 / 255, -0.51 / 255, -1000, 1000, np.nan]
    dx, dr = Dev(x), Dev(res)
    dxp, drec = Dev(nbytes=4 * n), Dev(nbytes=n)
    capi.check(L.qcnn_op_ppro_float(dx.p, dxp.p, n, None))
    capi.check(L.qcnn_op_apply_res_float(dx.p, dr.p, drec.p, n, None))
    xp_o = np.empty(n, np.float32)
    rec_o = np.empty(n, np.uint8)
    c_oracle.lib().orf_ppro(c_oracle._p(x), c_oracle._p(xp_o), n)
    c_oracle.lib().orf_apply_res(c_oracle._p(x), c_oracle._p(res), c_oracle._p(rec_o), n)
    assert np.array_equal(dxp.get(np.float32, n).view(np.uint32), xp_o.view(np.uint32))
    assert np.array_equal(drec.get(np.uint8, n), rec_o)
    for d in (dx, dr, dxp, drec):
        d.free()


def _check_rec(rec, rec_o):
    d = np.abs(rec.astype(int) - rec_o.astype(int))
    assert d.max() <= 1 and np.count_nonzero(d) <= max(1, rec.size // 1000), (d.max(), np.count_nonzero(d))


@pytest.mark.parametrize("qp,h,w", ((37, 240, 416), (22, 37, 53), (27, 8, 32), (32, 1, 1)))
def test_forward_float_against_oracle(qp, h, w):
    mb = synth.synth_float_model(qp)
    _, anc = synth.synth_frames(h, w, qp, 1)
    y = anc[0]
    rec_o, t = c_oracle.forward_float(mb, y, True)
    with qvrcnn(0, 1, 1, h, w) as net:
        net.load_float_para(mb)
        net.load_data(y)
        net.forward_float()
        net.synchronize()
        assert np.array_equal(net.tap(capi.TAP_F_X_PPRO).view(np.uint32), t["x_ppro"].view(np.uint32))
        for tap, key in ((capi.TAP_F_U_C1, "u1"), (capi.TAP_F_CONC1, "conc1"), (capi.TAP_F_CONC2, "conc2"),
                         (capi.TAP_F_U_C4, "u4")):
            g = net.tap(tap).reshape(t[key].shape)
            scale = max(1.0, float(np.abs(t[key]).max()))
            err = float(np.abs(g - t[key]).max())
            assert err <= 1e-5 * scale, f"{key}: max error {err} at scale {scale}"
        _check_rec(net.download(), rec_o)


def test_forward_float_1080p_crops():
    """Full-size frame: the summation order of a pixel does not depend on where its tile lies, and zero padding
    adds exact zeros, so the frame's interior equals -- bit for bit -- the engine's own result on a crop with a
    6-pixel margin; crops are also checked against the oracle within the tolerance."""
    h, w, m = 1080, 1920, 6
    mb = synth.synth_float_model(37)
    _, anc = synth.synth_frames(h, w, 37, 1)
    y = anc[0]
    with qvrcnn(0, 1, 1, h, w) as net:
        net.load_float_para(mb)
        net.load_data(y)
        net.forward_float()
        net.synchronize()
        rec = net.download()
        u4 = net.tap(capi.TAP_F_U_C4)[0]
    assert np.abs(rec.astype(int) - y.astype(int)).max() > 0
    for (y0, x0, ch, cw) in ((0, 0, 70, 90), (1010, 1830, 70, 90), (500, 1000, 64, 100), (3, 1850, 41, 70)):
        crop = np.ascontiguousarray(y[y0:y0 + ch, x0:x0 + cw])
        with qvrcnn(0, 1, 1, ch, cw) as net:
            net.load_float_para(mb)
            net.load_data(crop)
            net.forward_float()
            net.synchronize()
            rec_c = net.download()
            u4_c = net.tap(capi.TAP_F_U_C4)[0]
        # rows / columns of the crop whose 13 x 13 receptive field lies inside the crop or ends at a frame border
        r0 = 0 if y0 == 0 else m
        r1 = ch if y0 + ch == h else ch - m
        c0 = 0 if x0 == 0 else m
        c1 = cw if x0 + cw == w else cw - m
        assert np.array_equal(u4[y0 + r0:y0 + r1, x0 + c0:x0 + c1].view(np.uint32), u4_c[r0:r1, c0:c1].view(np.uint32))
        assert np.array_equal(rec[y0 + r0:y0 + r1, x0 + c0:x0 + c1], rec_c[r0:r1, c0:c1])
        rec_o, t = c_oracle.forward_float(mb, crop, True)
        assert np.abs(u4_c - t["u4"]).max() <= 1e-5 * max(1.0, float(np.abs(t["u4"]).max()))
        _check_rec(rec_c, rec_o)


def test_model_hwcn2nchw_file(tmp_path):
    """model_HWCN2NCHW (inference/qvrcnn.cu:464-491) file to file, then load the result from its path."""
    L = capi.load()
    nchw = synth.synth_float_model(37)
    f = np.frombuffer(nchw, np.float32)
    hwcn, off = bytearray(), 0
    for ks, cin, cout in zip(c_oracle.F_KS, c_oracle.F_CIN, c_oracle.COUT):
        n = ks * ks * cin * cout
        hwcn += np.ascontiguousarray(f[off:off + n].reshape(cout, cin, ks, ks).transpose(2, 3, 1, 0)).tobytes()
        off += n
        hwcn += f[off:off + cout].tobytes()
        off += cout
    src, dst = tmp_path / "vrcnn_hwcn_%d.data", tmp_path / "vrcnn_nchw_%d.data"
    (tmp_path / "vrcnn_hwcn_37.data").write_bytes(bytes(hwcn))
    capi.check(L.qcnn_model_hwcn2nchw(str(src).encode(), str(dst).encode(), 37))
    assert (tmp_path / "vrcnn_nchw_37.data").read_bytes() == nchw
    with qvrcnn(0, 1, 1, 8, 8) as net:
        net.load_float_para(str(tmp_path / "vrcnn_nchw_37.data"))
        with pytest.raises(capi.QcnnError):
            net.load_float_para(str(tmp_path / "missing.data"))
    with qvrcnn(0, 1, 1, 8, 8) as net:
        with pytest.raises(capi.QcnnError):
            net.forward_float()
