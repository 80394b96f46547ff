"""float32 CPU restatement of the reference's TensorFlow forward (training/model.py).

TEST / BASELINE INFRASTRUCTURE ONLY.  The reference graph is TensorFlow 1.x (tf.contrib,
tf.placeholder: training/model.py:5-6,30) and neither TensorFlow nor a checkpoint exists in this
image, so this file restates, with torch CPU conv2d, exactly what VRCNN.model_blu computes:

    images_norm = images/255 - 128/255                        training/model.py:32
    blu_l = clip(conv2d_SAME(prev, w_l) + b_l, 0, blu_ub[l])  training/model.py:96-107
    conv4 = conv2d_SAME(blu3, w4) + b4                        training/model.py:108-109
    pred  = conv4 + images_norm                               training/model.py:56
    out   = clip(pred*255 + 128, 0, 255)                      training/model.py:285,289
    psnr  on np.round(out)                                    training/utils.py:16-24

Weights are obtained by de-quantising an int8 static model with the convention of
training/quantization.py (w = w_int*stepw, b = b_int*stepw/ratio, blu_ub = blu_adj), so the float
network and the int8 network describe the same function up to quantisation error.
PARITY UNPINNED: the reference holds no output of this graph.
"""
from __future__ import annotations

import numpy as np


def dequantise(model, quant_rows):
    """model: formats.StaticModel; quant_rows: [[stepw, ratio, blu_adj, blu_q, mul, shift] x 6]."""
    ws, bs, blu = [], [], []
    for lp, row in zip(model.layers, quant_rows):
        stepw, ratio, blu_adj = float(row[0]), float(row[1]), float(row[2])
        ws.append(lp.w.astype(np.float32) * np.float32(stepw))            # [K][C][r][s] = OIHW
        bs.append(lp.b.astype(np.float32) * np.float32(stepw / ratio))
        blu.append(np.float32(blu_adj))
    return ws, bs, blu


def forward_float(ws, bs, blu, y: np.ndarray, threads: int | None = None) -> np.ndarray:
    """y uint8 [H][W] -> float32 [H][W] (pred*255+128, clipped to [0,255], not rounded)."""
    import torch
    import torch.nn.functional as F

    if threads:
        torch.set_num_threads(threads)
    with torch.no_grad():
        x = torch.from_numpy(y.astype(np.float32))[None, None] / 255 - 128 / 255
        tw = [torch.from_numpy(np.ascontiguousarray(w)) for w in ws]
        tb = [torch.from_numpy(np.ascontiguousarray(b)) for b in bs]

        def conv(inp, l):
            return F.conv2d(inp, tw[l], tb[l], padding=(tw[l].shape[-1] - 1) // 2)

        blu1 = conv(x, 0).clamp(0, float(blu[0]))
        blu2 = torch.cat([conv(blu1, 1).clamp(0, float(blu[1])), conv(blu1, 2).clamp(0, float(blu[2]))], 1)
        blu3 = torch.cat([conv(blu2, 3).clamp(0, float(blu[3])), conv(blu2, 4).clamp(0, float(blu[4]))], 1)
        pred = conv(blu3, 5) + x
        out = (pred * 255 + 128).clamp(0, 255)
    return out[0, 0].numpy()


def psnr(target: np.ndarray, ref: np.ndarray) -> float:
    """training/utils.py:16-24."""
    import math
    t = np.round(np.asarray(target, "f"))
    r = np.round(np.asarray(ref, "f"))
    d = (r - t).flatten("C")
    return 10 * math.log10(255 ** 2.0 / np.mean(d ** 2.0))
