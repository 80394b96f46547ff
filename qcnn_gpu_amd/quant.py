"""Quantisation tool chain of the static-quantised VRCNN and the int8 model exporter.

The reference derives the per-layer fixed-point parameters in training/quantization.py and dumps a
float32 checkpoint in training/model.py:318-338, but ships no program that turns the two into the
int8 model file its inference driver loads.  This module restates the parameter derivation (same
numbers, see tests/test_quant.py against vectors generated from the reference module) and adds the
missing exporter:

    float dump (HWCN float32 w, float32 b per layer)
        -> stepw per layer, adjust_quant -> rows [stepw, ratio, blu_adj, blu_q, mul, shift]
        -> int8 weights round(w / stepw), int32 biases round(b * ratio / stepw)
        -> HWCN int8 model file -> NCHW_VECT_C model file (inference/qvrcnn.cu:558-585)

Host-side, numpy only; nothing here runs on the inference path.
"""
from __future__ import annotations

import pickle
import struct
from typing import List, Sequence, Tuple

import numpy as np

from .formats import LAYERS, LayerParams, StaticModel

Row = List[float]  # [stepw, ratio, blu_adj, blu_q, mul, shift]

# initial clip thresholds per QP (training/quantization.py:66-75, the table quantNsave uses)
BLU_INIT = {
    22: (0.1111, 0.05, 0.05, 0.022, 0.022, 0),
    27: (0.294, 0.172, 0.172, 0.101, 0.101, 0),
    32: (0.316, 0.198, 0.198, 0.125, 0.125, 0),
    37: (0.349, 0.243, 0.243, 0.169, 0.169, 0),
}


def _first_fitting_shift(target: float, shifts: range, top_of, fits) -> Tuple[int, int]:
    """Walks the shifts upwards; for every shift whose numerator top_of(shift) exceeds `target` the multiplier is
    round(top / target), and the first (mul, shift) that `fits` is returned.  When none fits, the last multiplier
    tried goes with the largest shift (what the reference's loops leave behind when they fall through)."""
    mul = 0
    for shift in shifts:
        top = top_of(shift)
        if top > target:
            mul = round(top / target)
            if fits(top, mul, shift):
                return mul, shift
    return mul, shifts[-1]


def mul_shift(max_u: float) -> Tuple[int, int]:
    """(mul, shift) with max_u * mul / 2**shift in (127, 127.5): the requantiser that maps the clip threshold to 127
    (training/quantization.py:5-14)."""
    return _first_fitting_shift(max_u, range(1, 28), lambda k: 127.5 * 2 ** k, lambda top, mul, k: 127 < max_u * mul / 2 ** k < 127.5)


def mul_shift_f(ratio: float) -> Tuple[int, int]:
    """(mul, shift) with 2**shift / mul within 2 % of `ratio`, shift >= 10: the residual scaler of the last layer
    (training/quantization.py:15-24)."""
    return _first_fitting_shift(ratio, range(10, 28), lambda k: 2 ** k, lambda top, mul, k: abs(top / mul - ratio) < 0.02 * ratio)


# A row is [stepw, ratio, blu_adj, blu_q, mul, shift]; the helpers below name its fields.
_STEPW, _RATIO, _BLU_ADJ, _BLU_Q, _MUL, _SHIFT = range(6)


def _threshold_in_units(blu: float, ratio: float, stepw: float) -> int:
    """The float clip threshold in integer units of the layer's accumulator."""
    return round(blu * ratio / stepw)


def _threshold_applied(mul: int, shift: int, stepw: float, ratio: float) -> float:
    """The float threshold the integer clip really applies: the accumulator value that (mul, shift) map to 127."""
    return 127 * 2 ** shift / mul * stepw / ratio


def _gain(mul: int, shift: int, stepw: float) -> float:
    """Output units per float unit and unit of ratio of a requantiser working on weights of step stepw."""
    return mul / stepw / 2 ** shift


def _ratio_behind(ratio: float, row: Row) -> float:
    """Integer units per float unit of the activations behind a requantised layer."""
    return ratio / row[_STEPW] * row[_MUL] / 2 ** row[_SHIFT]


def quant_qfp_layer(ratio: float, stepw: float, blu: float) -> Row:
    """A layer with its own threshold (C1): the requantiser is fitted to the threshold, then the threshold is moved
    to where that requantiser clips (training/quantization.py:25-31)."""
    mul, shift = mul_shift(_threshold_in_units(blu, ratio, stepw))
    blu_adj = _threshold_applied(mul, shift, stepw, ratio)
    return [stepw, ratio, blu_adj, _threshold_in_units(blu_adj, ratio, stepw), mul, shift]


def quant_qfp_concat(ratio: float, stepw1: float, blu1: float, stepw2: float, blu2: float) -> List[Row]:
    """Two branches that are concatenated must leave their requantisers on the same scale: both take the larger
    threshold, and the weight step of the branch with the larger gain is coarsened until the gains
    mul / stepw / 2**shift are equal (training/quantization.py:32-49)."""
    blu = max(blu1, blu2)
    units = (_threshold_in_units(blu, ratio, stepw1), _threshold_in_units(blu, ratio, stepw2))
    (mul1, shift1), (mul2, shift2) = mul_shift(units[0]), mul_shift(units[1])
    if _gain(mul1, shift1, stepw1) > _gain(mul2, shift2, stepw2):
        stepw1 = stepw2 * 2 ** shift2 / mul2 * mul1 / 2 ** shift1
    else:
        stepw2 = stepw1 * 2 ** shift1 / mul1 * mul2 / 2 ** shift2
    return [[stepw1, ratio, _threshold_applied(mul1, shift1, stepw1, ratio), units[0], mul1, shift1],
            [stepw2, ratio, _threshold_applied(mul2, shift2, stepw2, ratio), units[1], mul2, shift2]]


def quant_qfp_last(ratio: float, stepw: float) -> Row:
    """The residual layer C4: its output is brought back to the 8-bit pixel scale (ratio 255) by
    (x * mul + round) >> shift in applyRes_y; the weight step is moved to where that scaler is exact
    (training/quantization.py:50-53)."""
    mul, shift = mul_shift_f(ratio / 255 / stepw)
    return [ratio * mul / 2 ** shift / 255, ratio, 0, 0, mul, shift]


def adjust_quant(stepw_in: Sequence[float], blu_in: Sequence[float]) -> List[Row]:
    """Rows of the six layers C1, C2_1, C2_2, C3_1, C3_2, C4 from the weight steps and the float clip thresholds.
    `ratio` (integer units per float unit of the activations, 255 for the input pixels) is chained through the
    requantisers: behind a concatenation both branches have the same ratio, the first one's is taken
    (training/quantization.py:55-64)."""
    rows: List[Row] = []
    ratio = 255
    for stage in ((0,), (1, 2), (3, 4)):  # a single layer, then two pairs of concatenated branches
        if len(stage) == 1:
            new = [quant_qfp_layer(ratio, stepw_in[stage[0]], blu_in[stage[0]])]
        else:
            a, b = stage
            new = quant_qfp_concat(ratio, stepw_in[a], blu_in[a], stepw_in[b], blu_in[b])
        rows += new
        ratio = _ratio_behind(ratio, new[0])
    rows.append(quant_qfp_last(ratio, stepw_in[5]))
    return rows


def stepw_from_weights(weights: Sequence[np.ndarray]) -> List[float]:
    """Initial weight step of each layer: the step that maps the extreme weight onto -128 / 127
    (training/quantization.py:77-83)."""
    out = []
    for w in weights:
        hi, lo = float(np.max(w)), float(np.min(w))
        out.append(hi / 127 if hi / 127 > -lo / 128 else -lo / 128)
    return out


# ---- float checkpoint dump -------------------------------------------------------------------
def float_dump_bytes(weights: Sequence[np.ndarray], biases: Sequence[np.ndarray]) -> bytes:
    """The file VRCNN.dump writes (training/model.py:325-337): per layer the float32 weights in
    TensorFlow's HWCN order [r][s][c][k], then the float32 biases."""
    out = bytearray()
    for l, (w, b) in enumerate(zip(weights, biases)):
        _, ks, cin, cout = LAYERS[l]
        assert w.shape == (ks, ks, cin, cout) and b.shape == (cout,)
        out += np.ascontiguousarray(w, "<f4").tobytes()
        out += np.ascontiguousarray(b, "<f4").tobytes()
    return bytes(out)


def read_float_dump(buf: bytes) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """Inverse of float_dump_bytes: ([w HWCN float32] * 6, [b float32] * 6)."""
    need = sum(4 * (k * k * ci * co + co) for _, k, ci, co in LAYERS)
    if len(buf) < need:
        raise ValueError(f"float dump too short: {len(buf)} < {need}")
    ws, bs, off = [], [], 0
    for _, ks, cin, cout in LAYERS:
        n = ks * ks * cin * cout
        ws.append(np.frombuffer(buf, "<f4", n, off).reshape(ks, ks, cin, cout).copy())
        off += 4 * n
        bs.append(np.frombuffer(buf, "<f4", cout, off).copy())
        off += 4 * cout
    return ws, bs


# ---- the exporter ----------------------------------------------------------------------------
def quantise_model(weights: Sequence[np.ndarray], biases: Sequence[np.ndarray], rows: Sequence[Sequence[float]]) -> StaticModel:
    """int8 model from float HWCN weights / biases and the six parameter rows: w_int = round(w / stepw)
    saturated to int8, b_int = round(b * ratio / stepw) (the integer parts of quant_w_b,
    training/quantization.py:100-106), blu / mul / shift = row[3:6]."""
    m = StaticModel()
    for l, (w, b) in enumerate(zip(weights, biases)):
        stepw, ratio, _, blu_q, mul, shift = rows[l]
        wq = np.clip(np.around(np.asarray(w, np.float64) / stepw), -128, 127).astype(np.int8)
        bq = np.around(np.asarray(b, np.float64) * ratio / stepw)
        if np.any(np.abs(bq) >= 2 ** 31):
            raise ValueError(f"layer {l}: bias does not fit int32")
        m.layers.append(LayerParams(np.ascontiguousarray(wq.transpose(3, 2, 0, 1)), bq.astype(np.int32), int(blu_q), int(mul),
                                    int(shift)))
    return m


def dequantise_model(m: StaticModel, rows: Sequence[Sequence[float]]):
    """Float view of an int8 model: w = w_int * stepw (HWCN), b = b_int * stepw / ratio, clip
    thresholds blu_adj -- what build_recon_model feeds model_blu (training/model.py:29-56)."""
    ws, bs, blu = [], [], []
    for l, lp in enumerate(m.layers):
        stepw, ratio, blu_adj = rows[l][0], rows[l][1], rows[l][2]
        ws.append((lp.w.astype(np.float64) * stepw).transpose(2, 3, 1, 0).astype(np.float32))
        bs.append((lp.b.astype(np.float64) * stepw / ratio).astype(np.float32))
        blu.append(float(blu_adj))
    return ws, bs, blu


def export(float_dump: bytes, qp: int, blu_init: Sequence[float] = None):
    """float dump -> (rows, StaticModel).  The weight steps of concatenated branches are adjusted by
    adjust_quant, so the weights are quantised with the adjusted rows, not with the initial steps."""
    ws, bs = read_float_dump(float_dump)
    rows = adjust_quant(stepw_from_weights(ws), BLU_INIT[qp] if blu_init is None else blu_init)
    return rows, quantise_model(ws, bs, rows)


def quant_params_pickle(rows: Sequence[Sequence[float]]) -> bytes:
    """quant_params%d.data as loadQpara reads it (training/quantization.py:118-130): a pickled list of
    six six-element lists (plain Python numbers)."""
    return pickle.dumps([[float(r[0]), float(r[1]), float(r[2]), int(r[3]), int(r[4]), int(r[5])] for r in rows], protocol=2)


def quant_params_cpp(rows: Sequence[Sequence[float]]) -> bytes:
    """quant_params_cpp_%d.data: six doubles per layer (training/quantization.py:92-96)."""
    return b"".join(struct.pack("<6d", *[float(x) for x in r]) for r in rows)
