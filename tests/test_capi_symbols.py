"""The C-ABI library loads and exports every symbol include/qcnn.h declares (no GPU calls)."""
import ctypes
import os
import re

from qcnn_gpu_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "qcnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qcnn_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = ctypes.CDLL(capi.LIB_PATH)
    names = _declared()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/qcnn.h but not exported"
    assert set(names) == set(capi.SYMBOLS)


def test_host_only_entry_points():
    lib = capi.load()
    assert lib.qcnn_version().startswith(b"qcnn-gfx950")
    # argument errors are reported through return codes + qcnn_last_error, not by aborting
    assert lib.qcnn_layer_params(None, 0, None, None, None) == -1
    assert b"bad layer" in lib.qcnn_last_error()
    assert lib.qcnn_load_static(None, b"x") == -1


def test_host_weight_relayout_matches_formula():
    import numpy as np
    lib = capi.load()
    h, w, c, n = 3, 3, 5, 2
    src = (np.arange(h * w * c * n) % 127).astype(np.int8)
    dst = np.full(n * 2 * h * w * 4, 99, np.int8)
    total = lib.qcnn_hwcn2nchw_vect_c(src.ctypes.data, h, w, c, n, dst.ctypes.data)
    assert total == dst.size
    d = dst.reshape(n, 2, h, w, 4)
    s = src.reshape(h, w, c, n)
    for ch in range(c):
        assert np.array_equal(d[:, ch >> 2, :, :, ch & 3], s[:, :, ch, :].transpose(2, 0, 1))
    assert (d[:, 1, :, :, 1:] == 0).all()       # pad lanes zero-filled (inference/mat.cu:108)
