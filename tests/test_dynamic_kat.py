"""Known answers for the dynamic-quantisation oracle (qvrcnn::forward, inference/qvrcnn.cu:82-167), derived
by hand from the reference's formulas, plus agreement of the two independent restatements (C and numpy)
and the committed golden cases.  The reference holds no vectors for this path ("parity unpinned")."""
import ctypes as C
import zlib

import numpy as np
import pytest

from oracle import c_oracle, np_dynamic
from qcnn_gpu_amd import formats, synth

BOFF = (0, 64, 96, 112, 128, 160)


def _ints(*v):
    return np.array(v, np.int32)


def test_quant_div_known_answers():
    # CHW2CHW_VECT_C, inference/mat.cu:209-225: round half away from zero, clip, char
    q = c_oracle.lib().orc_quant_div_one
    assert q(191.0, 3) == 64            # (191 + 1) / 3
    assert q(190.0, 3) == 63            # 191 / 3 = 63.67 -> truncated
    assert q(-5.0, 4) == -1             # (-5 - 2) / 4 = -1.75 -> truncated toward zero
    assert q(-6.0, 4) == -2             # exactly -2
    assert q(2.0, 4) == 1               # half rounds away from zero
    assert q(-2.0, 4) == -1
    assert q(1.0, 4) == 0
    assert q(1000.0, 3) == 127          # clip
    assert q(-1000.0, 3) == -128
    assert q(0.0, 7) == 0
    assert q(58843.0, 460) == 127       # a maximum lands on 127: (58843 + 230) / 460 = 128.4 -> clip
    assert q(58649.0, 460) == 127       # (58649 + 230) / 460 = 127.998 -> 127 by truncation
    u = np.arange(-70000, 70000, 37, dtype=np.float32)
    for d in (1, 2, 3, 123, 460):
        ref = np.array([q(float(v), d) for v in u], np.int8)
        assert np.array_equal(np_dynamic.quant_div(u, d), ref)


def test_step_known_answers():
    L = c_oracle.lib()
    # CovLayer::quantize_out, inference/cnn.cu:175: max / 128 + 1 in float, truncated
    assert L.orc_quantize_out_step(58843.0) == 460 == np_dynamic.quantize_out_step(np.float32(58843))
    assert L.orc_quantize_out_step(127.0) == 1
    assert L.orc_quantize_out_step(128.0) == 2
    assert L.orc_quantize_out_step(0.0) == 1
    # ConcatLayer::concat, inference/cnn.cu:295-309
    s1, s2 = C.c_int(), C.c_int()
    L.orc_concat_steps(1000.0, 300.0, 5, 3, C.byref(s1), C.byref(s2))
    # 1000/128+1 = 8, 300/128+1 = 3; 5*3 > 3*8 is false -> stepy2 = (3*8 + 2) / 5 = 5
    assert (s1.value, s2.value) == (8, 5) == np_dynamic.concat_steps(np.float32(1000), np.float32(300), 5, 3)
    L.orc_concat_steps(300.0, 1000.0, 5, 3, C.byref(s1), C.byref(s2))
    # 3 and 8; 5*8 > 3*3 -> stepy1 = (40 + 1) / 3 = 13
    assert (s1.value, s2.value) == (13, 8) == np_dynamic.concat_steps(np.float32(300), np.float32(1000), 5, 3)
    L.orc_concat_steps(100.0, 127.0, 7, 7, C.byref(s1), C.byref(s2))
    assert (s1.value, s2.value) == (1, 1)               # both maxima fit: steps stay 1
    L.orc_concat_steps(127.5, 0.0, 4, 4, C.byref(s1), C.byref(s2))
    assert (s1.value, s2.value) == (1, 1)               # 127.5 > 127 -> 127.5/128 + 1 = 1.99 -> 1


def test_insert_known_answers():
    # qvrcnn::insert_w keeps the list descending, insert_y ascending (inference/qvrcnn.cu:305-330)
    w, y = [0] * 4, [0] * 4
    for layer, v in enumerate((64, 181, 150, 300), 1):
        np_dynamic.insert_w(w, v, layer)
    assert w == [300, 181, 150, 64]
    for layer, v in enumerate((460, 221, 161), 1):
        np_dynamic.insert_y(y, v, layer)
    assert y == [161, 221, 460, 0]
    cw, cy = _ints(0, 0, 0, 0), _ints(0, 0, 0, 0)
    L = c_oracle.lib()
    for layer, v in enumerate((64, 181, 150, 300), 1):
        L.orc_insert_w(cw.ctypes.data, v, layer)
    for layer, v in enumerate((460, 221, 161), 1):
        L.orc_insert_y(cy.ctypes.data, v, layer)
    assert cw.tolist() == w and cy.tolist() == y
    # equal values go behind the ones already there
    w2 = [0] * 4
    for layer, v in enumerate((8, 8, 9), 1):
        np_dynamic.insert_w(w2, v, layer)
    assert w2 == [9, 8, 8, 0]


def test_adjust_basic_known_answers():
    # adjustBasic, inference/qvrcnn.cu:336-349
    stepw, stepy = _ints(7, 3, 0, 0), _ints(4, 5, 0, 0)
    b = np.array([10, -10, 0, 1, -1, 2.9], np.float32)
    out = np.empty_like(b)
    L = c_oracle.lib()
    L.orc_adjust_basic(stepw.ctypes.data, stepy.ctypes.data, b.ctypes.data, out.ctypes.data, b.size, 0)
    assert out.tolist() == [10, -10, 0, 1, -1, 2]          # n = 0: only the (long long) truncation
    L.orc_adjust_basic(stepw.ctypes.data, stepy.ctypes.data, b.ctypes.data, out.ctypes.data, b.size, 1)
    # 70 -> (70 + 2) / 4 = 18 ; -70 -> (-70 - 2) / 4 = -18 ; 7 -> 9/4 = 2 ; -7 -> -9/4 = -2 ; 14 -> 16/4 = 4
    assert out.tolist() == [18, -18, 0, 2, -2, 4]
    L.orc_adjust_basic(stepw.ctypes.data, stepy.ctypes.data, b.ctypes.data, out.ctypes.data, b.size, 2)
    # 210 -> 212/4 = 53 -> (53 + 2) / 5 = 11 ; 21 -> 23/4 = 5 -> 7/5 = 1 ; 42 -> 44/4 = 11 -> 13/5 = 2
    assert out.tolist() == [11, -11, 0, 1, -1, 2]
    for n in (0, 1, 2):
        L.orc_adjust_basic(stepw.ctypes.data, stepy.ctypes.data, b.ctypes.data, out.ctypes.data, b.size, n)
        assert np.array_equal(np_dynamic.adjust_basic(stepw, stepy, b, n), out)


def test_adjust_output_known_answers():
    # adjustOutput, inference/qvrcnn.cu:368-385, n = 2: temp = o * stepy[0], then / stepw[1], / stepw[0]
    stepw, stepy = _ints(5, 3, 0, 0), _ints(4, 0, 0, 0)
    o = np.array([10, -10, 0, 1, -1, 1000, -1000, 7.9], np.float32)
    out = np.empty(o.size, np.int8)
    c_oracle.lib().orc_adjust_output(stepw.ctypes.data, stepy.ctypes.data, o.ctypes.data, out.ctypes.data, 2, o.size)
    # 40 -> 41/3 = 13 -> 15/5 = 3 ; 4 -> 5/3 = 1 -> 3/5 = 0 ; 4000 -> 4001/3 = 1333 -> 1335/5 = 267 -> char 11
    # 28 -> 29/3 = 9 -> 11/5 = 2
    assert out.tolist() == [3, -3, 0, 0, 0, 11, -11, 2]
    assert np.array_equal(np_dynamic.adjust_output(stepw, stepy, o, 2), out)
    # adjustOutput_static, inference/qvrcnn.cu:386-397, with the constants of :157
    o2 = np.array([0, 232, 233, -232, -233, 465, 100000], np.float32)
    c_oracle.lib().orc_adjust_output_static(o2.ctypes.data, out.ctypes.data, 141, 16, o2.size)
    # 232*141 + 32768 = 65480 -> 0 ; 233*141 + 32768 = 65621 -> 1 ; -32712 + 32768 = 56 -> 0 ;
    # -32853 + 32768 = -85 -> -1 (arithmetic shift) ; 465*141 + 32768 = 98333 -> 1 ; 14132768 >> 16 = 215 -> char -41
    assert out[:o2.size].tolist() == [0, 0, 1, 0, -1, 1, -41]
    assert np.array_equal(np_dynamic.adjust_output_static(o2, 141, 16), out[:o2.size])


def test_mul_shift_and_apply_res_known_answers():
    L = c_oracle.lib()
    u = np.array([0, 100, -100, 5000, 0.9], np.float32)
    out = np.empty(u.size, np.int8)
    L.orc_mul_shift(u.ctypes.data, out.ctypes.data, u.size, 723, 16)   # bias = 32768 // 723 = 45
    # 45*723 >> 16 = 0 ; 145*723 = 104835 >> 16 = 1 ; -55*723 = -39765 >> 16 = -1 ; 5045*723 >> 16 = 55
    assert out.tolist() == [0, 1, -1, 55, 0]
    assert np.array_equal(np_dynamic.mul_shift(u, 723, 16), out)
    x = np.array([0, 3, 100, 250, 255, 128], np.uint8)
    res = np.array([-5, -3, 27, 10, 127, -128], np.int8)
    rec = np.empty_like(x)
    L.orc_apply_res(x.ctypes.data, res.ctypes.data, rec.ctypes.data, x.size)   # inference/cnn.cu:494-500
    assert rec.tolist() == [0, 0, 127, 255, 255, 0]
    assert np.array_equal(np_dynamic.apply_res(x, res), rec)


def test_find_max():
    u = np.array([[1.5, -7.25], [3.0, 7.0]], np.float32)
    assert c_oracle.lib().orc_find_max(u.ctypes.data, u.size) == 7.25 == float(np_dynamic.find_max(u))


def test_dynamic_model_format():
    m = synth.synth_dynamic_model(37)
    img = m.to_bytes()
    assert len(img) == formats.MODEL_DYNAMIC_BYTES == 59980 == c_oracle.lib().orc_dyn_model_bytes()
    # layer 0: step_w, then 64 x [1][5][5][4] weights with the single input channel in lane 0
    assert int.from_bytes(img[:4], "little", signed=True) == m.layers[0].step_w
    assert np.frombuffer(img, np.int8, 4, 4).tolist() == [int(m.layers[0].w[0, 0, 0, 0]), 0, 0, 0]
    back = formats.DynamicModel.from_bytes(img)
    for a, b in zip(m.layers, back.layers):
        assert a.step_w == b.step_w and np.array_equal(a.w, b.w) and np.array_equal(a.b, b.b)
    with pytest.raises(ValueError):
        formats.DynamicModel.from_bytes(img[:-1])


def _agree(t_np, t_c):
    for k in ("x_ppro", "v1", "conc1", "conc2", "v4"):
        assert np.array_equal(t_np[k], t_c[k]), k
    for l in range(6):
        assert np.array_equal(t_np["u"][l].view(np.uint32), t_c["u"][l].view(np.uint32)), f"u[{l}]"
        assert np.array_equal(t_np["b_adj"][l], t_c["b_adj"][BOFF[l]:BOFF[l] + t_np["b_adj"][l].size]), f"b_adj[{l}]"
    for k in ("max_u", "step_y", "stepw", "stepy"):
        assert list(t_np[k]) == list(t_c[k]), k


@pytest.mark.parametrize("qp", [22, 27, 32, 37])
@pytest.mark.parametrize("shape", [(1, 1), (5, 7), (19, 33), (40, 56)])
def test_c_and_numpy_oracles_agree_on_the_dynamic_pass(qp, shape):
    m = synth.synth_dynamic_model(qp)
    y = synth.synth_frames(*shape, qp, seed=3)[1][0]
    rec_n, t_n = np_dynamic.forward(m, y, True)
    rec_c, t_c = c_oracle.forward_dynamic(m.to_bytes(), y, True)
    assert np.array_equal(rec_n, rec_c)
    _agree(t_n, t_c)
    if shape == (40, 56):
        # every branch of the step logic is live on the synthetic models: the steps are well above 1, the
        # two branches of a concat disagree before they are reconciled, and the residual is not all zero
        assert min(t_n["step_y"][:5]) > 1 and t_n["step_y"][1] != t_n["step_y"][2]
        assert np.count_nonzero(t_n["v4"]) > t_n["v4"].size // 4


def test_small_maxima_keep_unit_steps():
    # weights so small that every maximum fits in 127: concat keeps step 1 and reconciles by step_w only
    m = synth.synth_dynamic_model(37, step_w=(1, 2, 3, 4, 4, 1))
    for lp in m.layers:
        lp.w = (lp.w.astype(np.int16) // 64).astype(np.int8)
        lp.b = (lp.b // 4096).astype(np.int32)
    y = synth.synth_frames(12, 17, 37)[1][0]
    rec_n, t_n = np_dynamic.forward(m, y, True)
    rec_c, t_c = c_oracle.forward_dynamic(m.to_bytes(), y, True)
    assert np.array_equal(rec_n, rec_c)
    _agree(t_n, t_c)
    assert t_n["step_y"][3] == t_n["step_y"][4] == 1


@pytest.mark.parametrize("qp", [22, 37])
def test_oracles_reproduce_committed_dynamic_golden(qp, golden_dir):
    case = np.load(f"{golden_dir}/dynamic_case_qp{qp}.npz")
    with open(f"{golden_dir}/dynamic_model_qp{qp}.data", "rb") as f:
        mb = f.read()
    rec, taps = c_oracle.forward_dynamic(mb, case["anchor"], True)
    assert np.array_equal(rec, case["rec"])
    assert taps["step_y"].tolist() == case["step_y"].tolist() and taps["max_u"].tolist() == case["max_u"].tolist()
    assert taps["stepw"].tolist() == case["stepw"].tolist() and taps["stepy"].tolist() == case["stepy"].tolist()
    assert zlib.crc32(taps["conc2"].tobytes()) == int(case["crc_conc2"])
    assert np.array_equal(np_dynamic.forward(formats.DynamicModel.from_bytes(mb), case["anchor"]), case["rec"])
