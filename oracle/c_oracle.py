"""ctypes loader for oracle/libqcnn_oracle.so (the C restatement in oracle/qcnn_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the qcnn_gpu_amd package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libqcnn_oracle.so")
COUT = (64, 32, 16, 16, 32, 1)


class _Taps(C.Structure):
    _fields_ = [("x_ppro", C.c_void_p), ("u", C.c_void_p * 6), ("v1", C.c_void_p), ("conc1", C.c_void_p),
                ("conc2", C.c_void_p)]


class _DynTaps(C.Structure):
    _fields_ = [("x_ppro", C.c_void_p), ("u", C.c_void_p * 6), ("v1", C.c_void_p), ("conc1", C.c_void_p),
                ("conc2", C.c_void_p), ("v4", C.c_void_p), ("b_adj", C.c_void_p), ("max_u", C.c_void_p),
                ("step_y", C.c_void_p), ("stepw", C.c_void_p), ("stepy", C.c_void_p)]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("qcnn_oracle.c", "qcnn_oracle_dyn.c", "qcnn_oracle_float.c")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libqcnn_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        L.orc_forward_blu.restype = C.c_int
        L.orc_forward_blu.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_psnr.restype = C.c_double
        L.orc_psnr.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.orc_sse.restype = C.c_uint64
        L.orc_sse.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.orc_requant_one.restype = C.c_int8
        L.orc_requant_one.argtypes = [C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_apply_res_one.restype = C.c_uint8
        L.orc_apply_res_one.argtypes = [C.c_uint8, C.c_float, C.c_int, C.c_int]
        L.orc_model_bytes.restype = C.c_long
        L.orc_model_hwcn_bytes.restype = C.c_long
        L.orc_model_hwcn2nchw_vect_c.restype = C.c_long
        L.orc_model_hwcn2nchw_vect_c.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long]
        L.orc_hwcn2nchw_vect_c.restype = C.c_long
        L.orc_hwcn2nchw_vect_c.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_model_params.restype = C.c_int
        L.orc_model_params.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        L.orc_conv_static.restype = None
        L.orc_conv_static.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_int, C.c_void_p]
        L.orc_quant_blu.restype = None
        L.orc_quant_blu.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_num_threads.restype = C.c_int
        vp, i, f, lg = C.c_void_p, C.c_int, C.c_float, C.c_long
        L.orc_dyn_model_bytes.restype = lg
        L.orc_find_max.restype = f
        L.orc_find_max.argtypes = [vp, lg]
        L.orc_quant_div_one.restype = C.c_int8
        L.orc_quant_div_one.argtypes = [f, i]
        L.orc_quant_div.restype = None
        L.orc_quant_div.argtypes = [vp, vp, lg, i, i]
        L.orc_mul_shift.restype = None
        L.orc_mul_shift.argtypes = [vp, vp, lg, i, i]
        L.orc_adjust_basic.restype = None
        L.orc_adjust_basic.argtypes = [vp, vp, vp, vp, i, i]
        L.orc_adjust_output.restype = None
        L.orc_adjust_output.argtypes = [vp, vp, vp, vp, i, lg]
        L.orc_adjust_output_static.restype = None
        L.orc_adjust_output_static.argtypes = [vp, vp, i, i, lg]
        L.orc_apply_res.restype = None
        L.orc_apply_res.argtypes = [vp, vp, vp, lg]
        L.orc_quantize_out_step.restype = i
        L.orc_quantize_out_step.argtypes = [f]
        L.orc_concat_steps.restype = None
        L.orc_concat_steps.argtypes = [f, f, i, i, C.POINTER(i), C.POINTER(i)]
        L.orc_insert_w.argtypes = [vp, i, i]
        L.orc_insert_y.argtypes = [vp, i, i]
        L.orc_forward_dynamic.restype = i
        L.orc_forward_dynamic.argtypes = [vp, lg, vp, i, i, vp, vp]
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orf_model_bytes.restype = lg
        L.orf_ppro.restype = None
        L.orf_ppro.argtypes = [vp, vp, lg]
        L.orf_conv.restype = None
        L.orf_conv.argtypes = [vp, vp, vp, vp, i, i, i, i, i, i, i]
        L.orf_apply_res_one.restype = C.c_uint8
        L.orf_apply_res_one.argtypes = [C.c_uint8, f]
        L.orf_apply_res.restype = None
        L.orf_apply_res.argtypes = [vp, vp, vp, lg]
        L.orf_forward.restype = i
        L.orf_forward.argtypes = [vp, lg, vp, i, i, vp, vp, vp, vp, vp, vp]
        L.orf_hwcn2nchw.restype = lg
        L.orf_hwcn2nchw.argtypes = [vp, i, i, i, i, vp]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def forward_blu(model_bytes: bytes, y: np.ndarray, want_taps: bool = False):
    """y uint8 [H][W] -> rec uint8 [H][W] (and the taps, in the reference's layouts)."""
    L = lib()
    y = np.ascontiguousarray(y, np.uint8)
    h, w = y.shape
    rec = np.empty((h, w), np.uint8)
    mb = np.frombuffer(model_bytes, np.uint8)
    taps = None
    tp = None
    if want_taps:
        taps = {
            "x_ppro": np.empty((1, h, w, 4), np.int8),
            "u": [np.empty((c, h, w), np.float32) for c in COUT],
            "v1": np.empty((16, h, w, 4), np.int8),
            "conc1": np.empty((12, h, w, 4), np.int8),
            "conc2": np.empty((12, h, w, 4), np.int8),
        }
        t = _Taps()
        t.x_ppro = taps["x_ppro"].ctypes.data
        for i in range(6):
            t.u[i] = taps["u"][i].ctypes.data
        t.v1 = taps["v1"].ctypes.data
        t.conc1 = taps["conc1"].ctypes.data
        t.conc2 = taps["conc2"].ctypes.data
        tp = C.byref(t)
    rc = L.orc_forward_blu(_p(mb), mb.size, _p(y), h, w, _p(rec), tp)
    if rc:
        raise ValueError(f"orc_forward_blu failed: {rc}")
    return (rec, taps) if want_taps else rec


def psnr(data: np.ndarray, ori: np.ndarray) -> float:
    d = np.ascontiguousarray(data, np.uint8)
    o = np.ascontiguousarray(ori, np.uint8)
    return float(lib().orc_psnr(_p(d), _p(o), d.size))


def sse(data: np.ndarray, ori: np.ndarray) -> int:
    d = np.ascontiguousarray(data, np.uint8)
    o = np.ascontiguousarray(ori, np.uint8)
    return int(lib().orc_sse(_p(d), _p(o), d.size))


def model_hwcn2nchw_vect_c(hwcn: bytes) -> bytes:
    L = lib()
    src = np.frombuffer(hwcn, np.uint8)
    dst = np.empty(L.orc_model_bytes(), np.uint8)
    n = L.orc_model_hwcn2nchw_vect_c(_p(src), src.size, _p(dst), dst.size)
    if n < 0:
        raise ValueError("orc_model_hwcn2nchw_vect_c failed")
    return dst[:n].tobytes()


def forward_dynamic(model_bytes: bytes, y: np.ndarray, want_taps: bool = False):
    """The dynamic-quantisation pass (oracle/qcnn_oracle_dyn.c): y uint8 [H][W] -> rec (and taps)."""
    L = lib()
    y = np.ascontiguousarray(y, np.uint8)
    h, w = y.shape
    rec = np.empty((h, w), np.uint8)
    mb = np.frombuffer(model_bytes, np.uint8)
    taps, tp = None, None
    if want_taps:
        taps = {
            "x_ppro": np.empty((1, h, w, 4), np.int8),
            "u": [np.empty((c, h, w), np.float32) for c in COUT],
            "v1": np.empty((16, h, w, 4), np.int8),
            "conc1": np.empty((12, h, w, 4), np.int8),
            "conc2": np.empty((12, h, w, 4), np.int8),
            "v4": np.empty((h, w), np.int8),
            "b_adj": np.empty(161, np.float32),
            "max_u": np.empty(6, np.int32), "step_y": np.empty(6, np.int32),
            "stepw": np.empty(4, np.int32), "stepy": np.empty(4, np.int32),
        }
        t = _DynTaps()
        for i in range(6):
            t.u[i] = taps["u"][i].ctypes.data
        for k in ("x_ppro", "v1", "conc1", "conc2", "v4", "b_adj", "max_u", "step_y", "stepw", "stepy"):
            setattr(t, k, taps[k].ctypes.data)
        tp = C.byref(t)
    rc = L.orc_forward_dynamic(_p(mb), mb.size, _p(y), h, w, _p(rec), tp)
    if rc:
        raise ValueError(f"orc_forward_dynamic failed: {rc}")
    return (rec, taps) if want_taps else rec


def forward_shipped(model_bytes: bytes, params, c1_step_y: int, y: np.ndarray):
    """qvrcnn::forward as its shipped text reads (oracle/qcnn_oracle_dyn.c orc_forward_shipped): params = 6 x (blu, mul,
    shift) the caller has put into the layer members, c1_step_y = the member C1.step_y.  Returns (rec, stepw, stepy,
    step_y)."""
    L = lib()
    y = np.ascontiguousarray(y, np.uint8)
    h, w = y.shape
    rec = np.empty((h, w), np.uint8)
    mb = np.frombuffer(model_bytes, np.uint8)
    q = np.ascontiguousarray(np.asarray(params, np.int32).reshape(18))
    stepw, stepy, step_y = np.zeros(4, np.int32), np.zeros(4, np.int32), np.zeros(6, np.int32)
    L.orc_forward_shipped.restype = C.c_int
    L.orc_forward_shipped.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.orc_forward_shipped(_p(mb), mb.size, _p(q), int(c1_step_y), _p(y), h, w, _p(rec), _p(stepw), _p(stepy), _p(step_y))
    if rc:
        raise ValueError(f"orc_forward_shipped failed: {rc}")
    return rec, stepw, stepy, step_y


# ---- float32 configuration (oracle/qcnn_oracle_float.c) ---------------------------------------------------
F_KS, F_CIN = (5, 3, 5, 3, 1, 3), (1, 64, 64, 48, 48, 48)


def conv_float(x: np.ndarray, w: np.ndarray, b: np.ndarray, relu: bool, acc32: bool = False) -> np.ndarray:
    """x float32 [cin][H][W], w float32 [cout][cin][ks][ks], b float32 [cout] -> u float32 [cout][H][W]."""
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    cout, cin, ks, _ = w.shape
    _, h, wd = x.shape
    u = np.empty((cout, h, wd), np.float32)
    lib().orf_conv(_p(x), _p(w), _p(b), _p(u), h, wd, cin, cout, ks, int(relu), int(acc32))
    return u


def forward_float(model_bytes: bytes, y: np.ndarray, want_taps: bool = False):
    """The float32 configuration's forward: y uint8 [H][W] -> rec uint8 [H][W] (and the float taps)."""
    L = lib()
    y = np.ascontiguousarray(y, np.uint8)
    h, w = y.shape
    rec = np.empty((h, w), np.uint8)
    mb = np.frombuffer(model_bytes, np.uint8)
    taps = None
    ptrs = [None] * 5
    if want_taps:
        taps = {"x_ppro": np.empty((h, w), np.float32), "u1": np.empty((64, h, w), np.float32),
                "conc1": np.empty((48, h, w), np.float32), "conc2": np.empty((48, h, w), np.float32),
                "u4": np.empty((h, w), np.float32)}
        ptrs = [_p(taps[k]) for k in ("x_ppro", "u1", "conc1", "conc2", "u4")]
    rc = L.orf_forward(_p(mb), mb.size, _p(y), h, w, _p(rec), *ptrs)
    if rc:
        raise ValueError(f"orf_forward failed: {rc}")
    return (rec, taps) if want_taps else rec
