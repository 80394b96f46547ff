"""On-disk formats of the reference, read and written without executing anything.

* static int8 model, NCHW_VECT_C order  -- read by CovLayer::load_static_para
  (inference/cnn.cu:90-112) in the layer order of inference/qvrcnn.cu:55-60
* static int8 model, HWCN order         -- input of layer_qfp_HWCN2NCHW_VECT_C
  (inference/qvrcnn.cu:535-557)
* quant_params%d.data                   -- pickle written by training/quantization.py:89-91
* quant_params_cpp_%d.data              -- 6 x struct.pack('6d') (training/quantization.py:93-96)
* recon_psnr.data / psnr_static_%d.data -- raw doubles appended by inference/kernel.cu:112-115
* log.txt record                        -- inference/kernel.cu:108-111
* planar YUV 4:2:0 8-bit, luma only     -- inference/yuv_data.cpp:15-42,113-128
"""
from __future__ import annotations

import math
import pickletools
import struct
import time
from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np

# Layer table of qvrcnn::qvrcnn (inference/qvrcnn.cu:11-18): name, ksize, inC, outC.
LAYERS = (
    ("C1", 5, 1, 64),
    ("C2_1", 3, 64, 32),
    ("C2_2", 5, 64, 16),
    ("C3_1", 3, 48, 16),
    ("C3_2", 1, 48, 32),
    ("C4", 3, 48, 1),
)
LAYER_NAMES = tuple(l[0] for l in LAYERS)
MACS_PER_PIXEL = sum(k * k * ci * co for _, k, ci, co in LAYERS)  # 54 512


def pad4(c: int) -> int:
    return (c + 3) // 4 * 4


def layer_wsize(l: int) -> int:
    """wSize of inference/cnn.cu:24."""
    _, ks, cin, cout = LAYERS[l]
    return ks * ks * pad4(cin) * cout


MODEL_BYTES = sum(layer_wsize(l) + 4 * LAYERS[l][3] + 12 for l in range(6))  # 60 028
MODEL_HWCN_BYTES = sum(k * k * ci * co + 4 * co + 12 for _, k, ci, co in LAYERS)  # 55 228


@dataclass
class LayerParams:
    """One layer in canonical (framework-neutral) form: w[k][c][r][s] int8, b[k] int32."""

    w: np.ndarray
    b: np.ndarray
    blu: int
    mul: int
    shift: int


@dataclass
class StaticModel:
    layers: List[LayerParams] = field(default_factory=list)

    # ---- NCHW_VECT_C file (what the inference program loads) -------------------------
    def to_nchw_vect_c_bytes(self) -> bytes:
        out = bytearray()
        for l, lp in enumerate(self.layers):
            _, ks, cin, cout = LAYERS[l]
            assert lp.w.shape == (cout, cin, ks, ks) and lp.w.dtype == np.int8
            wv = np.zeros((cout, pad4(cin) // 4, ks, ks, 4), np.int8)
            for c in range(cin):
                wv[:, c >> 2, :, :, c & 3] = lp.w[:, c]
            out += wv.tobytes()
            out += np.asarray(lp.b, "<i4").tobytes()
            out += struct.pack("<3i", lp.blu, lp.mul, lp.shift)
        assert len(out) == MODEL_BYTES
        return bytes(out)

    @staticmethod
    def from_nchw_vect_c_bytes(buf: bytes) -> "StaticModel":
        if len(buf) < MODEL_BYTES:
            raise ValueError(f"model image too short: {len(buf)} < {MODEL_BYTES}")
        m = StaticModel()
        off = 0
        for l in range(6):
            _, ks, cin, cout = LAYERS[l]
            n = layer_wsize(l)
            wv = np.frombuffer(buf, np.int8, n, off).reshape(cout, pad4(cin) // 4, ks, ks, 4)
            off += n
            w = np.empty((cout, cin, ks, ks), np.int8)
            for c in range(cin):
                w[:, c] = wv[:, c >> 2, :, :, c & 3]
            b = np.frombuffer(buf, "<i4", cout, off).astype(np.int32)
            off += 4 * cout
            blu, mul, shift = struct.unpack_from("<3i", buf, off)
            off += 12
            m.layers.append(LayerParams(w, b, blu, mul, shift))
        return m

    # ---- HWCN file (what the training side would export) -----------------------------
    def to_hwcn_bytes(self) -> bytes:
        out = bytearray()
        for l, lp in enumerate(self.layers):
            out += np.ascontiguousarray(lp.w.transpose(2, 3, 1, 0)).tobytes()  # [r][s][c][k]
            out += np.asarray(lp.b, "<i4").tobytes()
            out += struct.pack("<3i", lp.blu, lp.mul, lp.shift)
        assert len(out) == MODEL_HWCN_BYTES
        return bytes(out)

    @staticmethod
    def from_hwcn_bytes(buf: bytes) -> "StaticModel":
        if len(buf) < MODEL_HWCN_BYTES:
            raise ValueError("HWCN model image too short")
        m = StaticModel()
        off = 0
        for l in range(6):
            _, ks, cin, cout = LAYERS[l]
            n = ks * ks * cin * cout
            w = np.frombuffer(buf, np.int8, n, off).reshape(ks, ks, cin, cout).transpose(3, 2, 0, 1).copy()
            off += n
            b = np.frombuffer(buf, "<i4", cout, off).astype(np.int32)
            off += 4 * cout
            blu, mul, shift = struct.unpack_from("<3i", buf, off)
            off += 12
            m.layers.append(LayerParams(w, b, blu, mul, shift))
        return m


# ---------------------------------------------------------------------------------------
# Dynamic-quantisation model: what CovLayer::load_para reads (inference/cnn.cu:69-89), layer by layer in
# the order of qvrcnn::load_para (inference/qvrcnn.cu:38-43): int32 step_w, wSize int8 weights in
# NCHW_VECT_C order, outChannel int32 biases.  It is the output format of model_HWCN2NCHW_VECT_C
# (inference/qvrcnn.cu:398-443).
MODEL_DYNAMIC_BYTES = sum(4 + layer_wsize(l) + 4 * LAYERS[l][3] for l in range(6))  # 59 980


@dataclass
class DynamicLayerParams:
    w: np.ndarray      # int8 [k][c][r][s]
    b: np.ndarray      # int32 [k]
    step_w: int


@dataclass
class DynamicModel:
    layers: List[DynamicLayerParams] = field(default_factory=list)

    def to_bytes(self) -> bytes:
        out = bytearray()
        for l, lp in enumerate(self.layers):
            _, ks, cin, cout = LAYERS[l]
            assert lp.w.shape == (cout, cin, ks, ks) and lp.w.dtype == np.int8
            out += struct.pack("<i", lp.step_w)
            wv = np.zeros((cout, pad4(cin) // 4, ks, ks, 4), np.int8)
            for c in range(cin):
                wv[:, c >> 2, :, :, c & 3] = lp.w[:, c]
            out += wv.tobytes()
            out += np.asarray(lp.b, "<i4").tobytes()
        assert len(out) == MODEL_DYNAMIC_BYTES
        return bytes(out)

    @staticmethod
    def from_bytes(buf: bytes) -> "DynamicModel":
        if len(buf) < MODEL_DYNAMIC_BYTES:
            raise ValueError(f"dynamic model image too short: {len(buf)} < {MODEL_DYNAMIC_BYTES}")
        m = DynamicModel()
        off = 0
        for l in range(6):
            _, ks, cin, cout = LAYERS[l]
            (step_w,) = struct.unpack_from("<i", buf, off)
            off += 4
            n = layer_wsize(l)
            wv = np.frombuffer(buf, np.int8, n, off).reshape(cout, pad4(cin) // 4, ks, ks, 4)
            off += n
            w = np.empty((cout, cin, ks, ks), np.int8)
            for c in range(cin):
                w[:, c] = wv[:, c >> 2, :, :, c & 3]
            b = np.frombuffer(buf, "<i4", cout, off).astype(np.int32)
            off += 4 * cout
            m.layers.append(DynamicLayerParams(w, b, step_w))
        return m


# ---------------------------------------------------------------------------------------
# quant_params%d.data : a pickle of 6 lists of 6 numbers.  It is parsed here by hand from
# the opcode stream (pickletools.genops only tokenises); nothing from the file is imported,
# called or constructed.  The only REDUCE accepted is numpy's scalar reconstructor applied to
# an 8-byte little-endian float64 payload.
# ---------------------------------------------------------------------------------------
class _Mark:
    pass


class _Global:
    def __init__(self, name):
        self.name = name


class _Dtype:
    def __init__(self, code):
        self.code = code


def parse_quant_params_pickle(buf: bytes) -> List[List[float]]:
    stack: list = []
    memo: dict = {}
    for op, arg, _pos in pickletools.genops(buf):
        n = op.name
        if n in ("PROTO", "FRAME"):
            continue
        elif n == "EMPTY_LIST":
            stack.append([])
        elif n == "MARK":
            stack.append(_Mark)
        elif n in ("GLOBAL", "STACK_GLOBAL"):
            if n == "STACK_GLOBAL":
                name = stack.pop(); mod = stack.pop()
                arg = f"{mod} {name}"
            if arg not in ("numpy.core.multiarray scalar", "numpy._core.multiarray scalar", "numpy dtype"):
                raise ValueError(f"unexpected global in quant_params pickle: {arg!r}")
            stack.append(_Global(arg))
        elif n in ("BINUNICODE", "SHORT_BINUNICODE", "SHORT_BINBYTES", "BINBYTES", "BININT", "BININT1",
                   "BININT2", "BINFLOAT"):
            stack.append(arg)
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n in ("TUPLE1", "TUPLE2", "TUPLE3"):
            k = int(n[-1])
            items = stack[-k:]
            del stack[-k:]
            stack.append(tuple(items))
        elif n == "TUPLE":
            i = len(stack) - 1
            while stack[i] is not _Mark:
                i -= 1
            items = tuple(stack[i + 1:])
            del stack[i:]
            stack.append(items)
        elif n == "REDUCE":
            args = stack.pop()
            fn = stack.pop()
            if not isinstance(fn, _Global):
                raise ValueError("REDUCE on a non-global")
            if fn.name == "numpy dtype":
                stack.append(_Dtype(args[0]))
            else:  # numpy scalar(dtype, bytes)
                dt, payload = args
                if not isinstance(dt, _Dtype) or dt.code != "f8" or len(payload) != 8:
                    raise ValueError("only float64 numpy scalars are accepted")
                stack.append(struct.unpack("<d", payload)[0])
        elif n == "BUILD":
            stack.pop()  # dtype state tuple (byte order etc.): ignored, '<' assumed
        elif n in ("BINPUT", "LONG_BINPUT"):
            memo[arg] = stack[-1]
        elif n == "MEMOIZE":
            memo[len(memo)] = stack[-1]
        elif n in ("BINGET", "LONG_BINGET"):
            stack.append(memo[arg])
        elif n == "APPENDS":
            i = len(stack) - 1
            while stack[i] is not _Mark:
                i -= 1
            items = stack[i + 1:]
            del stack[i:]
            stack[-1].extend(items)
        elif n == "APPEND":
            v = stack.pop()
            stack[-1].append(v)
        elif n == "STOP":
            break
        else:
            raise ValueError(f"opcode {n} not allowed in quant_params pickle")
    top = stack[-1]
    if not (isinstance(top, list) and len(top) == 6 and all(isinstance(r, list) and len(r) == 6 for r in top)):
        raise ValueError("quant_params pickle is not a 6x6 list")
    return [[float(v) if not isinstance(v, int) else v for v in row] for row in top]


def read_quant_params(path: str) -> List[List[float]]:
    """[[stepw, ratio, blu_adj, blu_q, mul, shift] x 6] from quant_params%d.data."""
    with open(path, "rb") as f:
        return parse_quant_params_pickle(f.read())


def write_quant_params_cpp(path: str, params: Sequence[Sequence[float]]) -> None:
    """quant_params_cpp_%d.data: 6 x struct.pack('6d') (training/quantization.py:93-96)."""
    with open(path, "wb") as f:
        for row in params:
            f.write(struct.pack("6d", *[float(v) for v in row]))


def read_quant_params_cpp(path: str) -> List[List[float]]:
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) != 288:
        raise ValueError("quant_params_cpp file must be 288 bytes")
    return [list(struct.unpack_from("6d", buf, 48 * i)) for i in range(6)]


# ---- PSNR result files --------------------------------------------------------------------
def read_psnr_doubles(path: str) -> np.ndarray:
    """psnr_static_%d.data / recon_psnr.data: raw little-endian doubles (inference/kernel.cu:112-115)."""
    return np.fromfile(path, "<f8")


def append_psnr_double(path: str, value: float) -> None:
    with open(path, "ab") as f:
        f.write(struct.pack("<d", value))


def format_log_record(when: float, input_fn: str, frames: int, height: int, width: int, psnr1: float,
                      psnr2: float, micros: int) -> str:
    """The log.txt record of inference/kernel.cu:110 (ctime() supplies its own newline)."""
    return ("\nQVRCNN test date:%sdata:%s\nframes:%d\nheight:%d\nwidth:%d\nbefore net:PSNR=%f\n"
            "after quantized net:PSNR=%f\ntime:%dus\n") % (time.ctime(when) + "\n", input_fn, frames, height,
                                                          width, psnr1, psnr2, micros)


def format_stdout_record(psnr1: float, psnr2: float, micros: int) -> str:
    """The stdout line of inference/kernel.cu:107."""
    return "\nbefore net:PSNR=%.3f\nafter quantized net:PSNR=%.3f\ntime:%dus\n" % (psnr1, psnr2, micros)


# ---- YUV 4:2:0 8-bit, luma only ------------------------------------------------------------
def read_yuv420_luma(path: str, frames: int, height: int, width: int, start: int = 0) -> np.ndarray:
    """vrcnn_data::read_data (inference/yuv_data.cpp:32-38): per frame H*W luma bytes, then H*W/2
    chroma bytes are skipped.  Returns uint8 [frames][H][W]."""
    hw = height * width
    stride = hw + hw // 2
    out = np.empty((frames, height, width), np.uint8)
    with open(path, "rb") as f:
        for i in range(frames):
            f.seek((start + i) * stride)
            buf = f.read(hw)
            if len(buf) != hw:
                raise ValueError(f"{path}: short read at frame {start + i}")
            out[i] = np.frombuffer(buf, np.uint8).reshape(height, width)
    return out


def write_yuv420_luma(path: str, luma: np.ndarray, chroma_value: int = 0) -> None:
    """vrcnn_data::save_recon_as (inference/yuv_data.cpp:113-128): luma + H*W/2 bytes of 0."""
    luma = np.asarray(luma, np.uint8)
    if luma.ndim == 2:
        luma = luma[None]
    _, h, w = luma.shape
    uv = bytes([chroma_value]) * (h * w // 2)
    with open(path, "wb") as f:
        for fr in luma:
            f.write(np.ascontiguousarray(fr).tobytes())
            f.write(uv)


def psnr_from_sse(sse: int, n: int) -> float:
    """vrcnn_data::psnr (inference/yuv_data.cpp:93-95) from an integer sum of squared errors."""
    mse = float(sse) / float(n)
    if mse == 0.0:
        return float("inf")
    return 10 * math.log10(65025.0 / mse)  # the C library's log10, as the reference calls it
