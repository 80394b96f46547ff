"""numpy restatement of qvrcnn::forward_blu (inference/qvrcnn.cu:168-242).

TEST INFRASTRUCTURE ONLY (see oracle/qcnn_oracle.c for the rules); written independently of the
C oracle so that the two can pin each other.  PARITY UNPINNED BY THE REFERENCE: there are no
golden vectors for this path in the reference; the known answers live in tests/test_oracle_kat.py.

Works on the canonical model form (w[k][c][r][s], formats.StaticModel) and produces the same
taps, in the reference's layouts, as the C oracle.
"""
from __future__ import annotations

import numpy as np


def to_vect_c(chw: np.ndarray) -> np.ndarray:
    """[C][H][W] -> NCHW_VECT_C [C/4][H][W][4] (pad lanes zero), inference/mat.cu:139."""
    c, h, w = chw.shape
    cp = (c + 3) // 4 * 4
    out = np.zeros((cp // 4, h, w, 4), chw.dtype)
    for ch in range(c):
        out[ch >> 2, :, :, ch & 3] = chw[ch]
    return out


def from_vect_c(v: np.ndarray, c: int) -> np.ndarray:
    """NCHW_VECT_C [C/4][H][W][4] -> [C][H][W]."""
    return np.stack([v[ch >> 2, :, :, ch & 3] for ch in range(c)])


def ppro(x: np.ndarray) -> np.ndarray:
    """HW2HW_VECT_C_PPRO (inference/cnn.cu:445-453): int8(x - 128), [1][H][W]."""
    return (x.astype(np.int16) - 128).astype(np.int8)[None]


def conv_acc(inp: np.ndarray, w: np.ndarray) -> np.ndarray:
    """Exact integer cross-correlation, zero padding (ks-1)/2 (inference/cnn.cu:44-49).
    inp int8 [C][H][W], w int8 [K][C][ks][ks] -> int64 [K][H][W].  The contraction runs through
    float64 BLAS, which is exact here: every partial sum is an integer below 2^53."""
    k, c, ks, _ = w.shape
    _, h, wd = inp.shape
    p = (ks - 1) // 2
    xp = np.zeros((c, h + 2 * p, wd + 2 * p), np.float64)
    xp[:, p:p + h, p:p + wd] = inp
    acc = np.zeros((k, h * wd), np.float64)
    wf = w.astype(np.float64)
    for r in range(ks):
        for s in range(ks):
            acc += wf[:, :, r, s] @ xp[:, r:r + h, s:s + wd].reshape(c, h * wd)
    return np.rint(acc).astype(np.int64).reshape(k, h, wd)


def conv_static(inp: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """ConvForward_static (inference/cnn.cu:145-160): float32(acc) + float32(b), NCHW float32."""
    acc = conv_acc(inp, w).astype(np.int32)
    return acc.astype(np.float32) + b.astype(np.float32)[:, None, None]


def quant_blu(u: np.ndarray, blu: int, mul: int, shift: int) -> np.ndarray:
    """CHW2CHW_VECT_C_QUANT_BLU arithmetic (inference/mat.cu:268,286-291), layout left [C][H][W]."""
    bias = (1 << (shift - 1)) // mul
    t = np.where(u < 0, np.float32(0), u)  # keeps (int) conversion in range on the clipped arms
    t = np.where(u > np.float32(blu), np.float32(0), t)
    q = ((t.astype(np.int64) + bias) * mul) & 0xFFFFFFFF
    q = np.where(q >= 1 << 31, q - (1 << 32), q) >> shift  # int32 wrap + arithmetic shift
    q = q.astype(np.int64)
    q = np.where(u < 0, 0, q)
    q = np.where(u > np.float32(blu), 127, q)
    return (q & 0xFF).astype(np.uint8).view(np.int8)


def apply_res_y(x: np.ndarray, res: np.ndarray, mul: int, shift: int) -> np.ndarray:
    """applyRes_GPU_y (inference/cnn.cu:507-523)."""
    bias = 1 << (shift - 1)
    t = np.trunc(res.astype(np.float64)).astype(np.int64)
    t = (t * mul + bias) & 0xFFFFFFFF
    t = np.where(t >= 1 << 31, t - (1 << 32), t) >> shift
    rec = (x.astype(np.int64) + t) & 0xFFFF
    rec = np.where(rec >= 1 << 15, rec - (1 << 16), rec)  # stored to short
    return np.clip(rec, 0, 255).astype(np.uint8)


def forward_blu(model, y: np.ndarray, want_taps: bool = False):
    """model: formats.StaticModel; y: uint8 [H][W].  Returns rec (and the taps dict)."""
    L = model.layers
    h, w = y.shape
    x0 = ppro(y)
    u = [None] * 6
    u[0] = conv_static(x0, L[0].w, L[0].b)
    v1 = quant_blu(u[0], L[0].blu, L[0].mul, L[0].shift)
    u[1] = conv_static(v1, L[1].w, L[1].b)
    u[2] = conv_static(v1, L[2].w, L[2].b)
    conc1 = np.concatenate([quant_blu(u[1], L[1].blu, L[1].mul, L[1].shift),
                            quant_blu(u[2], L[2].blu, L[2].mul, L[2].shift)])
    u[3] = conv_static(conc1, L[3].w, L[3].b)
    u[4] = conv_static(conc1, L[4].w, L[4].b)
    conc2 = np.concatenate([quant_blu(u[3], L[3].blu, L[3].mul, L[3].shift),
                            quant_blu(u[4], L[4].blu, L[4].mul, L[4].shift)])
    u[5] = conv_static(conc2, L[5].w, L[5].b)
    rec = apply_res_y(y, u[5][0], L[5].mul, L[5].shift)
    if not want_taps:
        return rec
    taps = {
        "x_ppro": to_vect_c(x0),
        "u": u,
        "v1": to_vect_c(v1),
        "conc1": to_vect_c(conc1),
        "conc2": to_vect_c(conc2),
    }
    return rec, taps


def psnr(data: np.ndarray, ori: np.ndarray) -> float:
    """vrcnn_data::psnr (inference/yuv_data.cpp:87-97)."""
    d = data.astype(np.int64).ravel() - ori.astype(np.int64).ravel()
    mse = float((d * d).sum()) / d.size
    with np.errstate(divide="ignore"):
        return float(10 * np.log10(np.float64(65025.0) / np.float64(mse)))
