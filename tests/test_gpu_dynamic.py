"""GPU parity of the dynamic-quantisation pass (qvrcnn::forward, inference/qvrcnn.cu:82-167) through the C
ABI against the CPU oracle -- bit-exact on every buffer and on the step bookkeeping.

Tolerance: none.  Comparisons are np.array_equal on bytes, on float32 bit patterns, and == on integers.
"""
import ctypes as C
import zlib

import numpy as np
import pytest

from oracle import c_oracle, np_dynamic
from qcnn_gpu_amd import capi, formats, synth
from qcnn_gpu_amd.qvrcnn import qvrcnn

pytestmark = pytest.mark.gpu

U_TAPS = [capi.TAP_U_C1, capi.TAP_U_C2_1, capi.TAP_U_C2_2, capi.TAP_U_C3_1, capi.TAP_U_C3_2, capi.TAP_U_C4]


def _check_forward(net, mb, y, taps=True):
    """qcnn_forward (no float images, convolutions run twice) and, with taps, qcnn_forward_layered (every u
    materialised) against the oracle."""
    rec_o, t = c_oracle.forward_dynamic(mb, y, True)
    net.load_data(y)
    net.forward()
    net.synchronize()
    _check_state(net, t)
    for tp, key in ((capi.TAP_V_C1, "v1"), (capi.TAP_CONC1, "conc1"), (capi.TAP_CONC2, "conc2"), (capi.TAP_V_C4, "v4")):
        bad = np.argwhere(net.tap(tp) != t[key])
        assert bad.size == 0, f"{key} differs at {bad[:5].tolist()} ({len(bad)} bytes)"
    assert np.array_equal(net.tap(capi.TAP_U_C4).view(np.uint32), t["u"][5].view(np.uint32))
    rec = net.download()
    bad = np.argwhere(rec != rec_o)
    assert bad.size == 0, f"x_rec differs at {bad[:5].tolist()} ({len(bad)} pixels)"
    if not taps:
        return rec_o, t
    capi.check(capi.load().qcnn_dev_memset(net.x_rec_ptr, 0, rec.size))
    net.forward_layered()
    net.synchronize()
    _check_state(net, t)
    assert np.array_equal(net.tap(capi.TAP_X_PPRO), t["x_ppro"])
    for l, tp in enumerate(U_TAPS):
        assert np.array_equal(net.tap(tp).view(np.uint32), t["u"][l].view(np.uint32)), f"u[{l}]"
    assert np.array_equal(net.tap(capi.TAP_V_C1), t["v1"])
    assert np.array_equal(net.tap(capi.TAP_CONC1), t["conc1"])
    assert np.array_equal(net.tap(capi.TAP_CONC2), t["conc2"])
    assert np.array_equal(net.tap(capi.TAP_V_C4), t["v4"])
    assert np.array_equal(net.download(), rec_o)
    return rec_o, t


def _check_state(net, t):
    st = net.dynamic_state()
    assert list(st.max_u) == t["max_u"].tolist()
    assert list(st.step_y) == t["step_y"].tolist()
    assert list(st.stepw) == t["stepw"].tolist() and list(st.stepy) == t["stepy"].tolist()
    assert np.array_equal(np.array(st.b_adj, np.float32).view(np.uint32), t["b_adj"].view(np.uint32))


@pytest.mark.parametrize("qp", [22, 37])
def test_dynamic_golden_case(qp, golden_dir):
    case = np.load(f"{golden_dir}/dynamic_case_qp{qp}.npz")
    path = f"{golden_dir}/dynamic_model_qp{qp}.data"
    with open(path, "rb") as f:
        mb = f.read()
    y = case["anchor"]
    with qvrcnn(0, 1, 1, *y.shape) as net:
        net.load_para(path)
        rec_o, t = _check_forward(net, mb, y)
        assert np.array_equal(rec_o, case["rec"])
        assert zlib.crc32(net.tap(capi.TAP_CONC2).tobytes()) == int(case["crc_conc2"])
        assert list(net.dynamic_state().step_y) == case["step_y"].tolist()


@pytest.mark.parametrize("shape", [(1, 1), (2, 3), (5, 7), (19, 33), (64, 64), (63, 130), (240, 416)])
def test_dynamic_matches_oracle_on_ragged_sizes(shape):
    qp = (22, 27, 32, 37)[(shape[0] + shape[1]) % 4]
    m = synth.synth_dynamic_model(qp)
    mb = m.to_bytes()
    y = synth.synth_frames(*shape, qp, seed=5)[1][0]
    with qvrcnn(0, 1, 1, *shape) as net:
        net.load_para(mb)
        _check_forward(net, mb, y)
        # a second frame through the same engine: every maximum and step list starts afresh
        y2 = synth.synth_frames(*shape, qp, seed=6)[1][0]
        _check_forward(net, mb, y2, taps=False)


def test_dynamic_unit_steps_and_extreme_frames():
    m = synth.synth_dynamic_model(37, step_w=(1, 2, 3, 4, 4, 1))
    for lp in m.layers:
        lp.w = (lp.w.astype(np.int16) // 64).astype(np.int8)
        lp.b = (lp.b // 4096).astype(np.int32)
    mb = m.to_bytes()
    with qvrcnn(0, 1, 1, 12, 17) as net:
        net.load_para(mb)
        _, t = _check_forward(net, mb, synth.synth_frames(12, 17, 37)[1][0])
        assert t["step_y"][3] == t["step_y"][4] == 1
    mb = synth.synth_dynamic_model(27).to_bytes()
    with qvrcnn(0, 1, 1, 33, 47) as net:
        net.load_para(mb)
        for v in (0, 255, 128):
            _check_forward(net, mb, np.full((33, 47), v, np.uint8), taps=False)


def test_dynamic_full_hd_frame_bit_exact():
    qp = 37
    m = synth.synth_dynamic_model(qp)
    mb = m.to_bytes()
    y = synth.synth_frames(1080, 1920, qp)[1][0]
    rec_o, t = c_oracle.forward_dynamic(mb, y, True)
    with qvrcnn(0, 1, 1, 1080, 1920) as net:
        net.load_para(mb)
        net.load_data(y)
        net.forward()
        rec = net.download()
        st = net.dynamic_state()
        assert list(st.step_y) == t["step_y"].tolist() and list(st.max_u) == t["max_u"].tolist()
        assert np.array_equal(net.tap(capi.TAP_CONC2), t["conc2"])
        assert np.array_equal(rec, rec_o)


def test_static_and_dynamic_models_alternate_in_one_engine():
    sm = synth.synth_model(32).to_nchw_vect_c_bytes()
    dm = synth.synth_dynamic_model(32).to_bytes()
    y = synth.synth_frames(45, 70, 32)[1][0]
    with qvrcnn(0, 1, 1, 45, 70) as net:
        for _ in range(2):
            net.load_static_para(sm)
            net.load_data(y)
            net.forward_blu()
            assert np.array_equal(net.download(), c_oracle.forward_blu(sm, y))
            with pytest.raises(capi.QcnnError) as ei:
                net.forward()                    # the dynamic model is gone: CovLayer::w holds the static one
            assert ei.value.code == -5
            net.load_para(dm)
            _check_forward(net, dm, y, taps=False)
            with pytest.raises(capi.QcnnError) as ei:
                net.forward_blu()
            assert ei.value.code == -5


def test_dynamic_errors():
    with qvrcnn(0, 1, 1, 8, 8) as net:
        with pytest.raises(capi.QcnnError) as ei:
            net.forward()
        assert ei.value.code == -5
        with pytest.raises(capi.QcnnError) as ei:
            net.load_para("/nonexistent/model.data")
        assert ei.value.code == -2
        bad = bytearray(synth.synth_dynamic_model(37).to_bytes())
        bad[0:4] = (0).to_bytes(4, "little")      # C1.step_w = 0 would divide by zero in concat / adjustOutput
        with pytest.raises(capi.QcnnError) as ei:
            net.load_para(bytes(bad))
        assert ei.value.code == -3
        with pytest.raises(capi.QcnnError) as ei:
            net.load_para(bytes(bad[:5000]))
        assert ei.value.code == -3


def test_dynamic_layer_ops_on_raw_pointers():
    """qcnn_op_* of the dynamic path on caller-owned device buffers, each against the C oracle."""
    lib, L = capi.load(), c_oracle.lib()
    rng = np.random.default_rng(11)
    h, w, ch = 21, 35, 18                       # 18 channels: a ragged last vec4 group
    hw = h * w
    u = (rng.standard_normal((ch, h, w)) * 9000).astype(np.int32).astype(np.float32)
    u.flat[::97] = 0
    held = []

    def dev(a):
        p = C.c_void_p()
        capi.check(lib.qcnn_dev_alloc(C.byref(p), max(a.nbytes, 4)))
        capi.check(lib.qcnn_dev_h2d(p, a.ctypes.data, a.nbytes))
        held.append(p)
        return p

    def back(p, like):
        out = np.empty_like(like)
        capi.check(lib.qcnn_dev_sync())
        capi.check(lib.qcnn_dev_d2h(out.ctypes.data, p, out.nbytes))
        return out

    du = dev(u)
    # findMax
    mx = C.c_float()
    capi.check(lib.qcnn_op_find_max(du, dev(np.zeros(1, np.float32)), u.size, C.byref(mx), None))
    assert mx.value == L.orc_find_max(u.ctypes.data, u.size) == np.abs(u).max()
    # CHW2CHW_VECT_C for several divisors
    for d in (1, 2, 3, 77, 460):
        ref = np.zeros(((ch + 3) // 4, h, w, 4), np.int8)
        L.orc_quant_div(u.ctypes.data, ref.ctypes.data, hw, ch, d)
        dv = dev(np.zeros_like(ref))
        capi.check(lib.qcnn_op_quant_div(du, dv, hw, ch, d, None))
        assert np.array_equal(back(dv, ref), ref), f"quant_div divisor {d}"
    # mul_shift
    ref = np.empty(u.size, np.int8)
    L.orc_mul_shift(u.ctypes.data, ref.ctypes.data, u.size, 723, 16)
    dv = dev(np.zeros_like(ref))
    capi.check(lib.qcnn_op_mul_shift(du, dv, u.size, 723, 16, None))
    assert np.array_equal(back(dv, ref), ref)
    # activate
    du2 = dev(u)
    capi.check(lib.qcnn_op_activate(du2, u.size, None))
    assert np.array_equal(back(du2, u), np.maximum(u, 0))
    # adjustBasic / adjustOutput / adjustOutput_static
    stepw, stepy = np.array([300, 210, 181, 64], np.int32), np.array([161, 221, 460, 0], np.int32)
    b = (rng.standard_normal(48) * 4000).astype(np.int32).astype(np.float32)
    for n in range(4):
        ref = np.empty_like(b)
        L.orc_adjust_basic(stepw.ctypes.data, stepy.ctypes.data, b.ctypes.data, ref.ctypes.data, b.size, n)
        db = dev(np.zeros_like(b))
        capi.check(lib.qcnn_op_adjust_basic(stepw.ctypes.data, stepy.ctypes.data, dev(b), db, b.size, n, None))
        assert np.array_equal(back(db, ref), ref), f"adjust_basic n={n}"
    o = u[0].copy()
    for n in (1, 2, 4):
        ref = np.empty(o.size, np.int8)
        L.orc_adjust_output(stepw.ctypes.data, stepy.ctypes.data, o.ctypes.data, ref.ctypes.data, n, o.size)
        do = dev(np.zeros_like(ref))
        capi.check(lib.qcnn_op_adjust_output(stepw.ctypes.data, stepy.ctypes.data, dev(o), do, n, o.size, None))
        assert np.array_equal(back(do, ref), ref), f"adjust_output n={n}"
    # wide dividends: products beyond 2^51 take the 64-bit division, the later stages the reciprocal path; small
    # divisors and exact multiples sit on the truncation boundaries
    o_w = np.concatenate([rng.integers(-2 ** 19, 2 ** 19, 20000), np.arange(-3000, 3000), [0, 1, -1]]).astype(np.float32)
    for sw_, sy_, n in (((7, 3, 2, 1), (20000, 30000, 10000, 0), 4), ((1, 1, 1, 1), (1, 1, 1, 0), 4),
                        ((1000003, 65537, 4099, 17), (20000, 30000, 10000, 0), 4), ((12, 5, 0, 0), (6, 0, 0, 0), 2),
                        ((2, 0, 0, 0), (0, 0, 0, 0), 1)):
        sw_, sy_ = np.array(sw_, np.int32), np.array(sy_, np.int32)
        ref = np.empty(o_w.size, np.int8)
        L.orc_adjust_output(sw_.ctypes.data, sy_.ctypes.data, o_w.ctypes.data, ref.ctypes.data, n, o_w.size)
        do = dev(np.zeros_like(ref))
        capi.check(lib.qcnn_op_adjust_output(sw_.ctypes.data, sy_.ctypes.data, dev(o_w), do, n, o_w.size, None))
        assert np.array_equal(back(do, ref), ref), f"adjust_output wide {sw_.tolist()}"
    ref = np.empty(o.size, np.int8)
    L.orc_adjust_output_static(o.ctypes.data, ref.ctypes.data, 141, 16, o.size)
    do = dev(np.zeros_like(ref))
    capi.check(lib.qcnn_op_adjust_output_static(dev(o), do, 141, 16, o.size, None))
    assert np.array_equal(back(do, ref), ref)
    # applyRes
    x = rng.integers(0, 256, hw, dtype=np.uint8)
    res = rng.integers(-128, 128, hw, dtype=np.int8)
    ref = np.empty_like(x)
    L.orc_apply_res(x.ctypes.data, res.ctypes.data, ref.ctypes.data, hw)
    dr = dev(np.zeros_like(x))
    capi.check(lib.qcnn_op_apply_res(dev(x), dev(res), dr, hw, None))
    assert np.array_equal(back(dr, ref), ref)
    # argument checks
    assert lib.qcnn_op_quant_div(du, dv, hw, ch, 0, None) == -1
    zero = np.zeros(4, np.int32)
    assert lib.qcnn_op_adjust_output(zero.ctypes.data, stepy.ctypes.data, du, dv, 2, 4, None) == -1
    for p in held:
        capi.check(lib.qcnn_dev_free(p))


def test_concat_steps_host_helper():
    lib, L = capi.load(), c_oracle.lib()
    a, b, ra, rb = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    for m1, m2, s1, s2 in ((1000.0, 300.0, 5, 3), (300.0, 1000.0, 5, 3), (100.0, 127.0, 7, 7), (26781.0, 23364.0, 181, 150),
                           (127.5, 0.0, 4, 4), (16777215.0, 1.0, 1, 300)):
        capi.check(lib.qcnn_concat_steps(m1, m2, s1, s2, C.byref(a), C.byref(b)))
        L.orc_concat_steps(m1, m2, s1, s2, C.byref(ra), C.byref(rb))
        assert (a.value, b.value) == (ra.value, rb.value)
    assert lib.qcnn_concat_steps(1.0, 1.0, 0, 3, C.byref(a), C.byref(b)) == -1
