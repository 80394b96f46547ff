"""Deterministic synthetic inputs: Y frames and int8 static models.

The reference ships neither weights nor video (its model files and YUV sequences lived on the
author's disk, inference/kernel.cu:7), so every workload here is synthetic, built from integer
arithmetic only (SplitMix64) so that it is bit-identical on every machine.  The quantisation
triples (blu_q, mul, shift) are the reference's own, decoded from training/quant_params%d.data.
"""
from __future__ import annotations

import numpy as np

from .formats import LAYERS, DynamicLayerParams, DynamicModel, LayerParams, StaticModel

# (blu_q, mul, shift) per layer, decoded from training/quant_params{22,27,32,37}.data
# (columns 4-6 of each row written by training/quantization.py:31,49,53).
QUANT_TRIPLES = {
    22: ((3849, 271, 13), (4526, 115, 12), (5312, 49, 11), (18635, 7, 10), (4923, 53, 11), (0, 5, 24)),
    27: ((8390, 31, 11), (6600, 79, 12), (13622, 153, 14), (6983, 299, 14), (6020, 347, 14), (0, 1, 12)),
    32: ((10354, 6431, 19), (10629, 49, 12), (10523, 99, 13), (7426, 281, 14), (4526, 115, 12), (0, 13, 15)),
    37: ((11512, 723, 16), (10182, 205, 14), (11030, 189, 14), (13154, 635, 16), (7580, 551, 15), (0, 7, 13)),
}

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n outputs of the SplitMix64 generator started at `seed` (uint64 array)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _approx_gauss_q8(seed: int, n: int) -> np.ndarray:
    """Integer approximately N(0, 1) * 256: sum of eight uniform bytes, centred (int32)."""
    r = splitmix64(seed, n)
    s = np.zeros(n, np.int64)
    for k in range(8):
        s += ((r >> np.uint64(8 * k)) & np.uint64(0xFF)).astype(np.int64)
    # sum of 8 U{0..255}: mean 1020, std = sqrt(8*(256^2-1)/12) ~= 209.0 ; scale to std 256
    return ((s - 1020) * 256 // 209).astype(np.int32)


def synth_frames(height: int, width: int, qp: int = 37, frames: int = 1, seed: int = 0x51434E4E):
    """Returns (ori, anchor), uint8 [frames][H][W]: smooth structure + texture, and the same with
    blocky noise whose strength grows with QP (stands in for an HEVC-decoded anchor)."""
    yy, xx = np.mgrid[0:height, 0:width].astype(np.int64)
    ori = np.empty((frames, height, width), np.uint8)
    anc = np.empty((frames, height, width), np.uint8)
    amp = {22: 3, 27: 5, 32: 8, 37: 12}.get(qp, 8)
    for f in range(frames):
        s = (seed ^ (qp << 32) ^ f) & 0xFFFFFFFFFFFFFFFF
        # integer "sinusoid": triangle waves of incommensurate periods + ramps + 8x8 block texture
        def tri(v, period):
            t = v % period
            return np.abs(2 * t - period) * 255 // period
        base = (tri(xx + 3 * f, 97) * 3 + tri(yy + 5 * f, 61) * 3 + tri(xx + 2 * yy, 173) * 2) // 8
        blk = splitmix64(s ^ 0xB10C, ((height + 7) // 8) * ((width + 7) // 8)).reshape((height + 7) // 8, -1)
        blk = ((blk >> np.uint64(40)) & np.uint64(0x3F)).astype(np.int64) - 32
        tex = np.kron(blk, np.ones((8, 8), np.int64))[:height, :width]
        fine = (_approx_gauss_q8(s ^ 0xF17E, height * width).reshape(height, width).astype(np.int64) * 6) >> 8
        o = np.clip(base + tex + fine, 16, 235)
        nblk = _approx_gauss_q8(s ^ 0xA11C, ((height + 3) // 4) * ((width + 3) // 4)).reshape((height + 3) // 4, -1)
        noise = (np.kron(nblk.astype(np.int64), np.ones((4, 4), np.int64))[:height, :width] * amp) >> 9
        pix = (_approx_gauss_q8(s ^ 0x9015E, height * width).reshape(height, width).astype(np.int64) * amp) >> 9
        a = np.clip(o + noise + pix, 0, 255)
        ori[f] = o.astype(np.uint8)
        anc[f] = a.astype(np.uint8)
    return ori, anc


# Weight standard deviations (in int8 units, x1/16) per layer, tuned once so that on synth_frames
# input the accumulators spread over roughly [-blu_q/2, 1.5*blu_q]: all three arms of the
# requantiser (clip to 0, requantise, clip to 127) are exercised at every layer.
_SIGMA_Q4 = {
    22: (260, 60, 40, 330, 230, 40),
    27: (600, 80, 110, 120, 270, 40),
    32: (740, 130, 85, 125, 200, 40),
    37: (820, 125, 90, 230, 350, 40),
}


def synth_model(qp: int = 37, seed: int = 0x4D4F444C, sigma_q4=None, triples=None) -> StaticModel:
    """A random int8 static model with the reference's (blu_q, mul, shift) for `qp`."""
    triples = triples or QUANT_TRIPLES[qp]
    sig = sigma_q4 or _SIGMA_Q4[qp]
    m = StaticModel()
    for l, (_, ks, cin, cout) in enumerate(LAYERS):
        n = cout * cin * ks * ks
        g = _approx_gauss_q8(seed ^ (qp << 40) ^ (l << 8), n).astype(np.int64)
        w = np.clip((g * sig[l] + (1 << 11)) >> 12, -128, 127).astype(np.int8).reshape(cout, cin, ks, ks)
        blu, mul, shift = triples[l]
        span = max(blu, 2048) // 8
        r = splitmix64(seed ^ (qp << 40) ^ (l << 8) ^ 0xB1A5, cout)
        b = ((r >> np.uint64(20)) % np.uint64(2 * span + 1)).astype(np.int64) - span
        if l == 5:
            b = b // 16
        m.layers.append(LayerParams(w, b.astype(np.int32), int(blu), int(mul), int(shift)))
    return m


# Weight steps of the synthetic dynamic-quantisation models (CovLayer::step_w, inference/cnn.cu:78).  The
# two branches of a concat get different steps so that ConcatLayer::concat has to reconcile them, and
# the product is large enough for adjustOutput to leave residuals of a few grey levels.
_STEP_W = {
    22: (48, 150, 181, 160, 210, 260),
    27: (56, 181, 150, 190, 170, 280),
    32: (64, 170, 150, 210, 160, 300),
    37: (64, 181, 150, 210, 160, 300),
}


def synth_dynamic_model(qp: int = 37, step_w=None) -> DynamicModel:
    """The weights and biases of synth_model(qp) with a weight step per layer (dynamic model file)."""
    sw = step_w or _STEP_W[qp]
    m = synth_model(qp)
    return DynamicModel([DynamicLayerParams(lp.w, lp.b, int(s)) for lp, s in zip(m.layers, sw)])


def synth_float_model(qp: int = 37) -> bytes:
    """The float32 network that synth_model(qp) quantises, as an NCHW float model image (the file
    CovLayer::load_para reads in the float configuration, inference/cnn.cu:114-127): per layer float32
    weights [K][C][R][S] = w_int * stepw and float32 biases b_int * stepw / ratio."""
    from .quant_tables import QUANT_ROWS
    m = synth_model(qp)
    out = bytearray()
    for lp, row in zip(m.layers, QUANT_ROWS[qp]):
        stepw, ratio = float(row[0]), float(row[1])
        out += np.ascontiguousarray((lp.w.astype(np.float64) * stepw).astype("<f4")).tobytes()
        out += np.ascontiguousarray((lp.b.astype(np.float64) * stepw / ratio).astype("<f4")).tobytes()
    return bytes(out)


def adversarial_model(kind: str) -> StaticModel:
    """Edge-case models.
    'maxmag'   : every weight -128 / +127 so that |acc| crosses 2^24 in C2_2 (float-accumulator path).
    'identity' : C1 centre tap only, tiny ranges; accumulators sit exactly on 0 / blu / blu+1.
    'wide'     : full-magnitude weights on as many taps as keep sum|w| * 127 + |b| just below 2^23, with
                 the usual clip thresholds: accumulators run from far below zero to hundreds of times
                 the threshold, which is what the integer requantiser's saturating multiply must survive.
    'heavy'    : C1 and C2 at full magnitude on every tap with the usual requantisers and small biases: the C2_2
                 accumulators pass 2^24 on bright input (where the reference's float conversion rounds) and
                 -2^23 on the mixed-sign channels; C3 and C4 as trained.  Kernel A takes its clamped integer form.
    'c4big'    : the usual network with a C4 multiplier so large that applyRes_y's 32-bit product wraps (the reference
                 computes it in unsigned arithmetic): kernel B must replay the literal form, kernel A keeps integers.
    'balanced' : dense weights of both signs in equal numbers, as large as keep the sum of the positive and
                 the sum of the negative weights of a channel, times 127, below 2^23 each while sum|w| * 127
                 passes 2^23 in C2_1 and C2_2: activations are never negative, so the accumulators stay
                 24-bit operands although the magnitude bound alone cannot show it.
    """
    m = synth_model(37)
    if kind == "maxmag":
        for l, lp in enumerate(m.layers):
            sign = np.where((np.arange(lp.w.size) * 2654435761 >> 7) & 1, 127, -128).astype(np.int8)
            lp.w = sign.reshape(lp.w.shape)
            if l in (0, 1, 2):
                lp.w[::2] = 127  # half the output channels all-positive: huge positive accumulators
            lp.b = np.full(lp.b.shape, (1 << 24) + 3 if l == 2 else 12345, np.int32)
            if l < 5:
                lp.blu, lp.mul, lp.shift = (1 << 25), 1, 18
        return m
    if kind == "c4big":
        m.layers[5].mul, m.layers[5].shift = 3_000_017, 24
        return m
    if kind == "heavy":
        for l in (0, 1, 2):
            lp = m.layers[l]
            co = lp.w.shape[0]
            per = lp.w[0].size
            w = np.empty((co, per), np.int64)
            for k in range(co):
                i = np.arange(per)
                if l == 0:
                    w[k] = 127 if k % 4 != 3 else -128               # three in four C1 channels saturate on white input
                elif k % 2 == 0:
                    w[k] = 127                                       # all-positive channel
                else:
                    w[k] = np.where((i * 7 + k) % 20 < 3, 127, -128)  # 85 % negative
            lp.w = w.astype(np.int8).reshape(lp.w.shape)
            lp.b = ((np.arange(co) % 9 - 4) * 3000 + 17).astype(np.int32)
        return m
    if kind == "wide":
        for l, lp in enumerate(m.layers):
            co = lp.w.shape[0]
            per = lp.w[0].size
            in_max = 128 if l == 0 else 127
            n = min(per, ((1 << 23) - 70000) // (127 * in_max))  # nonzero taps per output channel
            w = np.zeros((co, per), np.int8)
            for k in range(co):
                idx = (np.arange(n) * 7 + k * 3) % per if per > n else np.arange(per)
                sgn = np.where(((np.arange(idx.size) + k) * 2654435761 >> 9) & 1, 127, -127)
                if k % 3 == 0:
                    sgn[:] = 127   # all-positive channel: far above the threshold on bright input
                w[k, idx] = sgn
            lp.w = w.reshape(lp.w.shape)
            lp.b = ((np.arange(co) % 7 - 3) * 9000).astype(np.int32)
        return m
    if kind == "balanced":
        for l, lp in enumerate(m.layers):
            co = lp.w.shape[0]
            per = lp.w[0].size
            # per-tap magnitude: sum|w| * 127 about 1.45 * 2^23 where the layer has taps enough for it
            base = min(127, int(1.45 * (1 << 23) / 127 / per))
            w = np.empty((co, per), np.int64)
            for k in range(co):
                i = np.arange(per)
                mag = np.maximum(base - (i + k) % 5, 1)
                sgn = np.where((i + k) & 1, 1, -1)            # equal numbers of both signs
                w[k] = mag * sgn
            lp.w = w.astype(np.int8).reshape(lp.w.shape)
            lp.b = ((np.arange(co) % 5 - 2) * 4000).astype(np.int32)
        return m
    if kind == "identity":
        for l, lp in enumerate(m.layers):
            lp.w[...] = 0
            lp.b[...] = 0
        c1 = m.layers[0]
        c1.w[:, 0, 2, 2] = np.arange(64) - 20  # acc = (x-128) * k
        c1.b[:] = np.arange(64) - 32
        c1.blu, c1.mul, c1.shift = 100, 127, 7  # bias = 64/127 = 0
        for l in (1, 2, 3, 4):
            lp = m.layers[l]
            ks = lp.w.shape[2]
            for k in range(lp.w.shape[0]):
                lp.w[k, k % lp.w.shape[1], ks // 2, ks // 2] = 1 + (k % 3)
            lp.b[:] = np.arange(lp.b.size) - 3
            lp.blu, lp.mul, lp.shift = 90, 181, 7
        c4 = m.layers[5]
        c4.w[0, :, 1, 1] = ((np.arange(48) % 5) - 2).astype(np.int8)
        c4.b[:] = -7
        c4.mul, c4.shift = 3, 2
        return m
    raise ValueError(kind)
