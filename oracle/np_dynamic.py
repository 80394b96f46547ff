"""numpy restatement of the dynamic-quantisation pass qvrcnn::forward (inference/qvrcnn.cu:82-167).

TEST INFRASTRUCTURE ONLY: imported by tests/, never by the shipped package.  PARITY UNPINNED BY THE
REFERENCE: it ships no vectors for this path; the known answers live in tests/test_dynamic_kat.py and
the C restatement (oracle/qcnn_oracle_dyn.c) is written independently so that the two pin each other.

`forward()` as shipped is caught mid-experiment: three of its calls read members that only
load_static_para fills (quantize_out_static at :100, concat_blu at :122/:141, adjustOutput_static(141, 16)
at :157), while the model it runs on comes from load_para, which never sets them.  The pass restated here
is the self-consistent one those lines replaced and that the source keeps next to them:
    C1.quantize_out(workspace)                      (:98,  inference/cnn.cu:169-178)
    Conc.concat(&a, &b, workspace)                  (:121/:140, inference/cnn.cu:285-320)
    adjustOutput(steps, C4.u, C4.v, layer, ...)     (:156, inference/qvrcnn.cu:368-385)
Everything else follows the active statements: adjustBasic before every convolution, no activation
after C1 (:96 is commented out), ReLU after C2_x / C3_x, insert_w / insert_y after every layer,
applyRes on the int8 residual.  save_steps (:163) appends a debug file and is not part of the result.
"""
from __future__ import annotations

import numpy as np

from . import np_oracle as npo

THRESHOLD = 127  # inference/mat.cuh:72


def cdiv(a: int, b: int) -> int:
    """C integer division (truncation toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b > 0) else -q


def find_max(u: np.ndarray) -> np.float32:
    """findMax (inference/mat.cu:28-96): the largest |u|."""
    return np.float32(np.abs(u).max())


def quant_div(u: np.ndarray, divisor: int) -> np.ndarray:
    """CHW2CHW_VECT_C arithmetic (inference/mat.cu:209-225), layout left [C][H][W]: float32
    (u -+ divisor/2) / divisor, clipped to [-128, 127], then truncated by the store to char."""
    u = u.astype(np.float32)
    d = np.float32(divisor)
    half = np.float32(divisor >> 1)
    neg = u < 0
    t = np.where(neg, (u - half) / d, (u + half) / d).astype(np.float32)
    t = np.where(neg, np.where(t < -128, np.float32(-128), t), np.where(t > 127, np.float32(127), t))
    return np.trunc(t).astype(np.int8)


def mul_shift(u: np.ndarray, mul: int, shift: int) -> np.ndarray:
    """mul_shift kernel (inference/mat.cu:248-261): no relayout, no clip, stored to char."""
    bias = (1 << (shift - 1)) // mul
    t = np.trunc(u.astype(np.float64)).astype(np.int64)
    q = ((t + bias) * mul) & 0xFFFFFFFF
    q = np.where(q >= 1 << 31, q - (1 << 32), q) >> shift
    return (q & 0xFF).astype(np.uint8).view(np.int8)


def adjust_basic(stepw, stepy, b: np.ndarray, n: int) -> np.ndarray:
    """adjustBasic (inference/qvrcnn.cu:336-349): bias rescaled by the first n weight steps and divided,
    rounding half away from zero, by the first n activation steps.  b float32 -> float32."""
    out = np.empty(len(b), np.float32)
    for i, bv in enumerate(b):
        temp = int(np.trunc(np.float64(bv)))
        for k in range(n):
            temp *= int(stepw[k])
        for k in range(n):
            sy = int(stepy[k])
            temp = cdiv(temp + (sy >> 1), sy) if temp > 0 else cdiv(temp - (sy >> 1), sy)
        out[i] = np.float32(temp)
    return out


def adjust_output(stepw, stepy, o: np.ndarray, n: int) -> np.ndarray:
    """adjustOutput (inference/qvrcnn.cu:368-385): o * stepy[0..n-2] / stepw[n-1..0], every division
    rounding half away from zero with the sign tested once, stored to char."""
    temp = np.trunc(o.astype(np.float64)).astype(np.int64)
    for k in range(n - 1):
        temp = temp * int(stepy[k])
    pos = temp > 0
    for k in range(n - 1, -1, -1):
        sw = int(stepw[k])
        h = sw >> 1
        up = (temp + h) // sw                    # operands positive on this arm: floor == truncation
        dn = -((h - temp) // sw)                 # temp <= 0 on this arm, so temp - h <= 0: truncation toward zero
        temp = np.where(pos, up, dn)             # the sign is tested once; neither arm can change it
    return (temp & 0xFF).astype(np.uint8).view(np.int8)


def adjust_output_static(o: np.ndarray, mul: int, shift: int) -> np.ndarray:
    """adjustOutput_static (inference/qvrcnn.cu:386-397): (int(o) * mul + (1 << shift-1)) >> shift."""
    t = np.trunc(o.astype(np.float64)).astype(np.int64)
    q = (t * mul + (1 << (shift - 1))) & 0xFFFFFFFF
    q = np.where(q >= 1 << 31, q - (1 << 32), q) >> shift
    return (q & 0xFF).astype(np.uint8).view(np.int8)


def apply_res(x: np.ndarray, res: np.ndarray) -> np.ndarray:
    """applyRes_GPU (inference/cnn.cu:487-502): (short)x + res, clamped to [0, 255]."""
    return np.clip(x.astype(np.int64) + res.astype(np.int64), 0, 255).astype(np.uint8)


def quantize_out_step(mx: np.float32) -> int:
    """step_y of CovLayer::quantize_out (inference/cnn.cu:175): float max / 128 + 1, truncated."""
    return int(np.float32(mx) / np.float32(THRESHOLD + 1) + np.float32(1))


def concat_steps(max1: np.float32, max2: np.float32, sw1: int, sw2: int):
    """Step selection of ConcatLayer::concat (inference/cnn.cu:295-309): one step per branch from its
    own maximum, then the finer branch is coarsened so that step_w * step_y agrees."""
    s1 = int(np.float32(max1) / np.float32(THRESHOLD + 1) + np.float32(1)) if max1 > THRESHOLD else 1
    s2 = int(np.float32(max2) / np.float32(THRESHOLD + 1) + np.float32(1)) if max2 > THRESHOLD else 1
    if sw1 * s2 > sw2 * s1:
        s1 = cdiv(sw1 * s2 + (sw2 >> 1), sw2)
    else:
        s2 = cdiv(sw2 * s1 + (sw1 >> 1), sw1)
    return s1, s2


def insert_w(stepw: list, value: int, layer: int) -> None:
    """qvrcnn::insert_w (inference/qvrcnn.cu:305-317): keeps stepw[0..layer-1] in descending order."""
    i = 0
    while i < layer - 1:
        if stepw[i] < value:
            for j in range(layer - 1, i, -1):
                stepw[j] = stepw[j - 1]
            stepw[i] = value
            return
        i += 1
    stepw[i] = value


def insert_y(stepy: list, value: int, layer: int) -> None:
    """qvrcnn::insert_y (inference/qvrcnn.cu:318-330): keeps stepy[0..layer-1] in ascending order."""
    i = 0
    while i < layer - 1:
        if stepy[i] > value:
            for j in range(layer - 1, i, -1):
                stepy[j] = stepy[j - 1]
            stepy[i] = value
            return
        i += 1
    stepy[i] = value


def conv_dynamic(inp: np.ndarray, w: np.ndarray, b_adj: np.ndarray, relu: bool) -> np.ndarray:
    """CovLayer::ConvForward (inference/cnn.cu:129-144) [+ CovLayer::activate, :161-166]."""
    u = npo.conv_acc(inp, w).astype(np.int32).astype(np.float32) + b_adj.astype(np.float32)[:, None, None]
    return np.maximum(u, np.float32(0)) if relu else u


def forward(model, y: np.ndarray, want_taps: bool = False):
    """model: formats.DynamicModel; y uint8 [H][W].  Returns rec (and taps / bookkeeping)."""
    L = model.layers
    stepw, stepy = [0] * 4, [0] * 4
    bf = [lp.b.astype(np.float32) for lp in L]
    b_adj, u, max_u, step_y = [None] * 6, [None] * 6, [0] * 6, [0] * 6
    x0 = npo.ppro(y)
    # layer 1
    b_adj[0] = adjust_basic(stepw, stepy, bf[0], 0)
    u[0] = conv_dynamic(x0, L[0].w, b_adj[0], False)
    mx = find_max(u[0])
    max_u[0] = int(mx)
    step_y[0] = quantize_out_step(mx)
    v1 = quant_div(u[0], step_y[0])
    insert_w(stepw, L[0].step_w, 1)
    insert_y(stepy, step_y[0], 1)
    # layers 2 and 3: two branches + concat
    prev, concs = v1, []
    for layer, (a, b) in ((2, (1, 2)), (3, (3, 4))):
        for l in (a, b):
            b_adj[l] = adjust_basic(stepw, stepy, bf[l], layer - 1)
        for l in (a, b):
            u[l] = conv_dynamic(prev, L[l].w, b_adj[l], True)
        m1, m2 = find_max(u[a]), find_max(u[b])
        max_u[a], max_u[b] = int(m1), int(m2)
        step_y[a], step_y[b] = concat_steps(m1, m2, L[a].step_w, L[b].step_w)
        prev = np.concatenate([quant_div(u[a], step_y[a]), quant_div(u[b], step_y[b])])
        concs.append(prev)
        insert_w(stepw, L[a].step_w, layer)
        insert_y(stepy, step_y[a], layer)
    # layer 4
    b_adj[5] = adjust_basic(stepw, stepy, bf[5], 3)
    u[5] = conv_dynamic(prev, L[5].w, b_adj[5], False)
    insert_w(stepw, L[5].step_w, 4)
    v4 = adjust_output(stepw, stepy, u[5][0], 4)
    rec = apply_res(y, v4)
    if not want_taps:
        return rec
    taps = {
        "x_ppro": npo.to_vect_c(x0), "u": u, "v1": npo.to_vect_c(v1), "conc1": npo.to_vect_c(concs[0]),
        "conc2": npo.to_vect_c(concs[1]), "v4": v4, "b_adj": b_adj, "max_u": max_u, "step_y": step_y,
        "stepw": list(stepw), "stepy": list(stepy),
    }
    return rec, taps
