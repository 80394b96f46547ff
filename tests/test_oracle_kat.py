"""Known answers for the oracle, derived by hand from the reference's formulas, plus agreement of
the two independent restatements (C and numpy).  The reference holds no golden vectors for this
path ("parity unpinned"): these are what pins the oracle."""
import os
import zlib

import numpy as np
import pytest

from oracle import c_oracle, np_oracle
from qcnn_gpu_amd import formats, synth


def test_requant_known_answers():
    # CHW2CHW_VECT_C_QUANT_BLU, inference/mat.cu:268,286-291.  QP37 C1: blu 11512, mul 723, shift 16
    # => bias = (1<<15)//723 = 45
    L = c_oracle.lib()
    q = lambda u, blu=11512, mul=723, sh=16: L.orc_requant_one(float(u), blu, mul, sh)
    assert q(-1) == 0
    assert q(0) == (45 * 723) >> 16 == 0
    assert q(5000) == (5045 * 723) >> 16 == 55
    assert q(11512) == (11557 * 723) >> 16 == 127
    assert q(11513) == 127            # clip arm, strict '>'
    assert q(-0.5) == 0               # '< 0' arm on a non-integer
    assert q(0.75) == 0               # (int) truncation, then (0+45)*723>>16
    # QP22 C3_1: blu 18635, mul 7, shift 10 => bias 73
    assert q(18635, 18635, 7, 10) == (18708 * 7) >> 10 == 127
    # numpy restatement agrees on a sweep
    u = np.arange(-300, 12000, 7, dtype=np.float32)
    got = np_oracle.quant_blu(u, 11512, 723, 16)
    ref = np.array([q(v) for v in u], np.int8)
    assert np.array_equal(got, ref)


def test_requant_stores_to_char():
    # a triple whose top value exceeds 127 wraps when stored to char, as the reference's store does
    L = c_oracle.lib()
    v = ((1000 + 1) * 1) >> 1  # bias = (1<<0)//1 = 1
    assert L.orc_requant_one(1000.0, 1000, 1, 1) == np.array([v & 0xFF], np.uint8).view(np.int8)[0]


def test_residual_known_answers():
    # applyRes_GPU_y, inference/cnn.cu:512-520.  QP37 C4: mul 7, shift 13.
    L = c_oracle.lib()
    r = lambda x, t: L.orc_apply_res_one(x, float(t), 7, 13)
    assert r(100, 100) == 100       # (700+4096)>>13 = 0
    assert r(100, 585) == 100       # 4095+4096 = 8191 -> 0
    assert r(100, 586) == 101
    assert r(100, 1000) == 101
    assert r(100, -585) == 100      # (-4095+4096)>>13 = 0
    assert r(100, -586) == 99       # arithmetic shift: floor
    assert r(100, -1000) == 99
    assert r(250, 10 * 1171) == 255  # +10 -> clamp high
    assert r(3, -10 * 1171) == 0     # -10 -> clamp low
    assert r(100, 585.9) == 100      # (int) truncates toward zero
    assert r(100, -586.9) == 99
    x = np.array([[0, 3, 100, 250, 255]], np.uint8)
    res = np.array([[-5000.0, -11710.0, 586.0, 11710.0, 40000.0]], np.float32)
    ref = np.array([[L.orc_apply_res_one(int(a), float(b), 7, 13) for a, b in zip(x[0], res[0])]], np.uint8)
    assert np.array_equal(np_oracle.apply_res_y(x, res, 7, 13), ref)


def test_psnr_known_answers():
    ori = np.full((4, 6), 100, np.uint8)
    assert c_oracle.psnr(ori, ori) == float("inf")           # mse = 0 (inference/yuv_data.cpp:95)
    assert abs(c_oracle.psnr(ori + 1, ori) - 48.130803608679104) < 1e-12   # 10*log10(65025)
    assert c_oracle.psnr(ori + 1, ori) == np_oracle.psnr(ori + 1, ori)
    assert c_oracle.sse(ori + 3, ori) == 9 * 24
    assert formats.psnr_from_sse(24, 24) == c_oracle.psnr(ori + 1, ori)


def test_weight_relayout_offsets():
    # HWCN2NCHW_VECT_C_CPU, inference/mat.cu:116: C1 file offset of w[k][0][r][s][lane 0] = k*100 + (r*5+s)*4
    m = synth.synth_model(37)
    img = m.to_nchw_vect_c_bytes()
    w1 = m.layers[0].w
    for k, r, s in [(0, 0, 0), (5, 2, 3), (63, 4, 4)]:
        off = k * 100 + (r * 5 + s) * 4
        assert np.frombuffer(img, np.int8, 1, off)[0] == w1[k, 0, r, s]
        assert img[off + 1:off + 4] == b"\0\0\0"          # pad lanes are zero
    assert len(img) == 60028 == formats.MODEL_BYTES == c_oracle.lib().orc_model_bytes()
    assert formats.MODEL_HWCN_BYTES == 55228 == c_oracle.lib().orc_model_hwcn_bytes()
    assert formats.MACS_PER_PIXEL == 54512


def test_hwcn_conversion_matches_between_implementations():
    m = synth.synth_model(27)
    hwcn = m.to_hwcn_bytes()
    assert c_oracle.model_hwcn2nchw_vect_c(hwcn) == m.to_nchw_vect_c_bytes()
    m2 = formats.StaticModel.from_hwcn_bytes(hwcn)
    assert m2.to_nchw_vect_c_bytes() == m.to_nchw_vect_c_bytes()


def test_centre_tap_identity_model_by_hand():
    # 'identity' model: C1 = centre tap only: u1[k] = (x-128)*(k-20) + (k-32); blu 100, mul 127, shift 7
    m = synth.adversarial_model("identity")
    y = np.array([[128, 129, 127, 255, 0, 138]], np.uint8)
    rec, taps = np_oracle.forward_blu(m, y, True)
    k = np.arange(64)[:, None, None]
    xs = y.astype(np.int64)[None] - 128
    u1 = xs * (k - 20) + (k - 32)
    assert np.array_equal(taps["u"][0], u1.astype(np.float32))
    v1 = np.where(u1 > 100, 127, np.where(u1 < 0, 0, (u1 * 127) >> 7))
    assert np.array_equal(np_oracle.from_vect_c(taps["v1"], 64), v1.astype(np.int8))
    rec_c, taps_c = c_oracle.forward_blu(m.to_nchw_vect_c_bytes(), y, True)
    assert np.array_equal(rec, rec_c)
    assert np.array_equal(taps["u"][5], taps_c["u"][5])


@pytest.mark.parametrize("qp", [22, 27, 32, 37])
@pytest.mark.parametrize("shape", [(1, 1), (5, 7), (16, 32), (19, 33), (40, 56)])
def test_c_and_numpy_oracles_agree(qp, shape):
    m = synth.synth_model(qp)
    _, anc = synth.synth_frames(shape[0], shape[1], qp, seed=qp * 1000 + shape[0])
    rec_n, taps_n = np_oracle.forward_blu(m, anc[0], True)
    rec_c, taps_c = c_oracle.forward_blu(m.to_nchw_vect_c_bytes(), anc[0], True)
    assert np.array_equal(rec_n, rec_c)
    for k in ("x_ppro", "v1", "conc1", "conc2"):
        assert np.array_equal(taps_n[k], taps_c[k]), k
    for l in range(6):
        assert np.array_equal(taps_n["u"][l], taps_c["u"][l]), l


@pytest.mark.parametrize("kind", ["maxmag", "identity", "heavy", "c4big"])
def test_oracles_agree_on_adversarial_models(kind):
    m = synth.adversarial_model(kind)
    for frame in (np.zeros((9, 20), np.uint8), np.full((9, 20), 255, np.uint8),
                  synth.synth_frames(12, 24, 37)[1][0]):
        rec_n, taps_n = np_oracle.forward_blu(m, frame, True)
        rec_c, taps_c = c_oracle.forward_blu(m.to_nchw_vect_c_bytes(), frame, True)
        assert np.array_equal(rec_n, rec_c)
        for l in range(6):
            assert np.array_equal(taps_n["u"][l], taps_c["u"][l]), l
    if kind == "maxmag":
        # the point of this model: C2_2 accumulators beyond 2^24, where float32 rounding matters
        assert np.abs(taps_c["u"][2]).max() > 2 ** 24
    if kind == "heavy":
        # ... and of this one: the same on a white frame with the usual requantisers (every such value clips to 127)
        _, taps_w = c_oracle.forward_blu(m.to_nchw_vect_c_bytes(), np.full((9, 20), 255, np.uint8), True)
        assert taps_w["u"][2].max() > 2 ** 24 and taps_w["u"][2].min() < -(2 ** 23)


@pytest.mark.parametrize("qp", [22, 27, 32, 37])
def test_oracles_reproduce_committed_golden(qp, golden_dir):
    case = np.load(f"{golden_dir}/synth_case_qp{qp}.npz")
    with open(f"{golden_dir}/synth_model_qp{qp}.data", "rb") as f:
        mb = f.read()
    assert mb == synth.synth_model(qp).to_nchw_vect_c_bytes()      # generator is deterministic
    ori, anc = synth.synth_frames(40, 56, qp)
    assert np.array_equal(ori[0], case["ori"]) and np.array_equal(anc[0], case["anchor"])
    rec, taps = c_oracle.forward_blu(mb, case["anchor"], True)
    assert np.array_equal(rec, case["rec"])
    assert np.array_equal(taps["conc1"], case["conc1"])
    crcs = dict(zip(case["crc_names"].tolist(), case["crc_values"].tolist()))
    for k in ("x_ppro", "v1", "conc1", "conc2"):
        assert zlib.crc32(taps[k].tobytes()) == crcs[k], k
    for l in range(6):
        assert zlib.crc32(taps["u"][l].tobytes()) == crcs[f"u{l}"], l
    assert c_oracle.psnr(rec, case["ori"]) == float(case["psnr_after"])
    m = formats.StaticModel.from_nchw_vect_c_bytes(mb)
    assert np.array_equal(np_oracle.forward_blu(m, case["anchor"]), case["rec"])


def test_psnr_against_reference_vectors():
    """vrcnn_data::psnr (inference/yuv_data.cpp:87-97: 10*log10(65025/mse)) as restated by the C oracle, by
    the package's host twin and from the device-side integer SSE, against values produced by the reference's
    own training/PSNR.py (20*log10(255/rmse); tools/make_psnr_golden.py).  The two formulas are the same
    number up to the last bits of a double: tolerance 1e-12 relative."""
    import json
    from qcnn_gpu_amd import formats
    path = os.path.join(os.path.dirname(__file__), "golden", "psnr_vectors.json")
    for c in json.load(open(path))["cases"]:
        rng = np.random.default_rng(c["seed"])
        h, w = c["shape"]
        a = rng.integers(0, 256, (h, w)).astype(np.uint8)
        b = np.clip(a.astype(np.int64) + rng.integers(-c["noise"], c["noise"] + 1, (h, w)), 0, 255).astype(np.uint8)
        if np.array_equal(a, b):
            b[0, 0] ^= 1
        want = c["psnr"]
        assert abs(c_oracle.psnr(b, a) - want) <= 1e-12 * abs(want)
        assert abs(formats.psnr_from_sse(c_oracle.sse(b, a), a.size) - want) <= 1e-12 * abs(want)
