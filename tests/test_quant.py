"""Quantisation tool chain (qcnn_gpu_amd/quant.py) against vectors produced by the reference's own
training/quantization.py (tests/golden/quantization_vectors.json, tools/make_quant_golden.py), and the
int8 exporter end to end.  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import c_oracle, float_model
from qcnn_gpu_amd import formats, quant, quant_tables, synth
from qcnn_gpu_amd.formats import LAYERS


@pytest.fixture(scope="module")
def vectors(golden_dir):
    with open(os.path.join(golden_dir, "quantization_vectors.json")) as f:
        return json.load(f)


def _same(a, b):
    if isinstance(a, (list, tuple)):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            _same(x, y)
    else:
        assert a == b, (a, b)          # the same float operations in the same order: identical doubles


def test_mul_shift_vectors(vectors):
    for arg, want in vectors["mul_shift"]:
        assert list(quant.mul_shift(arg)) == want
    for arg, want in vectors["mul_shift_f"]:
        assert list(quant.mul_shift_f(arg)) == want


def test_layer_rows_vectors(vectors):
    for args, want in vectors["quant_qfp_layer"]:
        _same(quant.quant_qfp_layer(*args), want)
    for args, want in vectors["quant_qfp_concat"]:
        _same(quant.quant_qfp_concat(*args), want)
    for args, want in vectors["quant_qfp_last"]:
        _same(quant.quant_qfp_last(*args), want)


def test_adjust_quant_vectors(vectors):
    for (stepw, blu), want in vectors["adjust_quant"]:
        _same(quant.adjust_quant(stepw, blu), want)


@pytest.mark.parametrize("qp", [22, 27, 32, 37])
def test_shipped_rows_are_a_fixed_point_of_the_chain(qp):
    """The rows the reference ships obey the ratio chain adjust_quant builds, and requantising the
    stored threshold gives 127 (the property the kernels' integer clip relies on)."""
    rows = quant_tables.QUANT_ROWS[qp]
    ratio = 255.0
    for l in (0, 1, 3):
        assert rows[l][1] == pytest.approx(ratio, rel=1e-12)
        stepw, _, _, _, mul, shift = rows[l]
        ratio = ratio / stepw * mul / 2 ** shift
    assert rows[5][1] == pytest.approx(ratio, rel=1e-12)
    for l in range(5):
        blu_q, mul, shift = rows[l][3:6]
        bias = (1 << (shift - 1)) // mul
        assert ((blu_q + bias) * mul) >> shift == 127


def _random_float_net(seed):
    rng = np.random.default_rng(seed)
    ws, bs = [], []
    for _, ks, cin, cout in LAYERS:
        ws.append((rng.standard_normal((ks, ks, cin, cout)) * 0.5 / np.sqrt(ks * ks * cin)).astype(np.float32))
        bs.append((rng.standard_normal(cout) * 0.01).astype(np.float32))
    return ws, bs


def test_float_dump_roundtrip():
    ws, bs = _random_float_net(1)
    buf = quant.float_dump_bytes(ws, bs)
    assert len(buf) == sum(4 * (k * k * ci * co + co) for _, k, ci, co in LAYERS)
    ws2, bs2 = quant.read_float_dump(buf)
    for a, b in zip(ws + bs, ws2 + bs2):
        assert np.array_equal(a, b)
    with pytest.raises(ValueError):
        quant.read_float_dump(buf[:-1])


@pytest.mark.parametrize("qp", [22, 37])
def test_export_float_dump_to_int8_model(qp, tmp_path):
    ws, bs = _random_float_net(qp)
    rows, model = quant.export(quant.float_dump_bytes(ws, bs), qp)
    # the rows satisfy the kernels' load-time requirements
    for l in range(5):
        blu_q, mul, shift = rows[l][3:6]
        assert 0 < mul < (1 << 23) and 1 <= shift <= 31
        assert ((blu_q + (1 << (shift - 1)) // mul) * mul) >> shift == 127
    # both branches of a concat leave their requantiser on one scale
    for a, b in ((1, 2), (3, 4)):
        sa = rows[a][4] / rows[a][0] / 2 ** rows[a][5]
        sb = rows[b][4] / rows[b][0] / 2 ** rows[b][5]
        assert sa == pytest.approx(sb, rel=1e-12)
    # weights: round(w / stepw), within half a step, saturating only at the int8 range
    for l, lp in enumerate(model.layers):
        w_hwcn = lp.w.transpose(2, 3, 1, 0).astype(np.float64) * rows[l][0]
        inside = np.abs(ws[l] / rows[l][0]) < 127
        assert np.max(np.abs(w_hwcn - ws[l])[inside]) <= 0.5 * rows[l][0] * (1 + 1e-9)
        assert np.array_equal(lp.b, np.around(bs[l].astype(np.float64) * rows[l][1] / rows[l][0]).astype(np.int32))
    # files: HWCN int8 -> NCHW_VECT_C through the reference's conversion restated in the oracle
    hwcn = model.to_hwcn_bytes()
    assert c_oracle.model_hwcn2nchw_vect_c(hwcn) == model.to_nchw_vect_c_bytes()
    # parameter files: the pickle is what loadQpara reads, and the hand parser reads it too
    p = tmp_path / f"quant_params{qp}.data"
    p.write_bytes(quant.quant_params_pickle(rows))
    back = formats.read_quant_params(str(p))
    _same([list(map(float, r)) for r in back], [list(map(float, r)) for r in rows])
    pc = tmp_path / f"quant_params_cpp_{qp}.data"
    pc.write_bytes(quant.quant_params_cpp(rows))
    assert formats.read_quant_params_cpp(str(pc)) == [list(map(float, r)) for r in rows]
    # the exported int8 network run by the integer oracle tracks the float network it came from
    y = synth.synth_frames(64, 64, qp, seed=3)[1][0]
    rec_int = c_oracle.forward_blu(model.to_nchw_vect_c_bytes(), y)
    fw, fb, blu = quant.dequantise_model(model, rows)                       # HWCN, as TensorFlow holds them
    fw = [np.ascontiguousarray(w.transpose(3, 2, 0, 1)) for w in fw]        # -> OIHW for the torch restatement
    rec_f = float_model.forward_float(fw, fb, [np.float32(b) for b in blu], y)
    d = np.abs(rec_int.astype(np.float64) - rec_f.astype(np.float64))
    assert d.mean() < 0.6 and d.max() <= 3.5
