"""Python view of the reference's `class qvrcnn` (inference/qvrcnn.cuh:25-59) over the C ABI.

Method names and call order follow the reference driver (inference/kernel.cu:86-96):
    net = qvrcnn(gpu_num, batch, channel, height, width)
    net.load_static_para(filename)
    net.load_data(frame)          # host uint8 [H][W]
    net.forward_blu()
    net.synchronize()             # cudaDeviceSynchronize() in the reference
    rec = net.download()          # the cudaMemcpy from I1.x_rec
The C++ twin of this class (include/qvrcnn.h) is what a C++ caller of the reference would use.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .formats import LAYERS

_TAP_SHAPES = {
    capi.TAP_X_PPRO: (np.int8, lambda h, w: (1, h, w, 4)),
    capi.TAP_U_C1: (np.float32, lambda h, w: (64, h, w)),
    capi.TAP_U_C2_1: (np.float32, lambda h, w: (32, h, w)),
    capi.TAP_U_C2_2: (np.float32, lambda h, w: (16, h, w)),
    capi.TAP_U_C3_1: (np.float32, lambda h, w: (16, h, w)),
    capi.TAP_U_C3_2: (np.float32, lambda h, w: (32, h, w)),
    capi.TAP_U_C4: (np.float32, lambda h, w: (1, h, w)),
    capi.TAP_V_C1: (np.int8, lambda h, w: (16, h, w, 4)),
    capi.TAP_CONC1: (np.int8, lambda h, w: (12, h, w, 4)),
    capi.TAP_CONC2: (np.int8, lambda h, w: (12, h, w, 4)),
    capi.TAP_X_REC: (np.uint8, lambda h, w: (h, w)),
    capi.TAP_FUSED_CONC1: (np.int8, lambda h, w: (12, h, w, 4)),
    capi.TAP_V_C4: (np.int8, lambda h, w: (h, w)),
    capi.TAP_F_X_PPRO: (np.float32, lambda h, w: (h, w)),
    capi.TAP_F_U_C1: (np.float32, lambda h, w: (64, h, w)),
    capi.TAP_F_CONC1: (np.float32, lambda h, w: (48, h, w)),
    capi.TAP_F_CONC2: (np.float32, lambda h, w: (48, h, w)),
    capi.TAP_F_U_C4: (np.float32, lambda h, w: (1, h, w)),
}


class qvrcnn:
    def __init__(self, gpu_num: int, batch: int, channel: int, height: int, width: int):
        self._lib = capi.load()
        self._h = C.c_void_p()
        capi.check(self._lib.qcnn_create(gpu_num, batch, channel, height, width, C.byref(self._h)))
        self.batch, self.channel, self.height, self.width = batch, channel, height, width

    # -- reference API ---------------------------------------------------------------------
    def load_static_para(self, filename) -> int:
        if isinstance(filename, (bytes, bytearray, memoryview)) and len(filename) > 4096:
            buf = bytes(filename)
            capi.check(self._lib.qcnn_load_static_mem(self._h, buf, len(buf)))
        else:
            name = filename if isinstance(filename, bytes) else str(filename).encode()
            capi.check(self._lib.qcnn_load_static(self._h, name))
        return 0

    def load_data(self, frame: np.ndarray) -> int:
        frame = np.ascontiguousarray(frame, np.uint8)
        if frame.size != self.height * self.width:
            raise ValueError("frame must hold channel*height*width bytes")
        capi.check(self._lib.qcnn_upload(self._h, frame.ctypes.data_as(C.c_void_p)))
        return 0

    def forward_blu(self) -> int:
        capi.check(self._lib.qcnn_forward_blu(self._h))
        return 0

    def load_para(self, filename) -> int:
        """qvrcnn::load_para (inference/qvrcnn.cu:30-46): the dynamic-quantisation model."""
        if isinstance(filename, (bytes, bytearray, memoryview)) and len(filename) > 4096:
            buf = bytes(filename)
            capi.check(self._lib.qcnn_load_para_mem(self._h, buf, len(buf)))
        else:
            name = filename if isinstance(filename, bytes) else str(filename).encode()
            capi.check(self._lib.qcnn_load_para(self._h, name))
        return 0

    def forward(self) -> int:
        """qvrcnn::forward (inference/qvrcnn.cu:82-167), the dynamic-quantisation pass."""
        capi.check(self._lib.qcnn_forward(self._h))
        return 0

    def forward_layered(self) -> int:
        """The dynamic pass with every float32 u materialised (tap mode)."""
        capi.check(self._lib.qcnn_forward_layered(self._h))
        return 0

    def dynamic_state(self) -> "capi.DynamicState":
        st = capi.DynamicState()
        capi.check(self._lib.qcnn_dynamic_state(self._h, C.byref(st)))
        return st

    def load_float_para(self, filename) -> int:
        """qvrcnn::load_para of the float32 configuration (inference/cnn.cu:114-127): the NCHW float model."""
        if isinstance(filename, (bytes, bytearray, memoryview)) and len(filename) > 4096:
            buf = bytes(filename)
            capi.check(self._lib.qcnn_load_float_mem(self._h, buf, len(buf)))
        else:
            name = filename if isinstance(filename, bytes) else str(filename).encode()
            capi.check(self._lib.qcnn_load_float(self._h, name))
        return 0

    def forward_float(self) -> int:
        """qvrcnn::forward of the float32 configuration (inference/qvrcnn.cu:244-285)."""
        capi.check(self._lib.qcnn_forward_float(self._h))
        return 0

    # -- additions ---------------------------------------------------------------------------
    def forward_blu_layered(self) -> int:
        capi.check(self._lib.qcnn_forward_blu_layered(self._h))
        return 0

    def forward_blu_batch(self, dev_y: int, dev_rec: int, nframes: int, stride: int = 0) -> int:
        capi.check(self._lib.qcnn_forward_blu_batch(self._h, C.c_void_p(dev_y), C.c_void_p(dev_rec), nframes,
                                                    stride or self.height * self.width))
        return 0

    def synchronize(self) -> None:
        capi.check(self._lib.qcnn_sync(self._h))

    def download(self) -> np.ndarray:
        out = np.empty((self.height, self.width), np.uint8)
        capi.check(self._lib.qcnn_download(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def tap(self, which: int) -> np.ndarray:
        dt, shp = _TAP_SHAPES[which]
        out = np.empty(shp(self.height, self.width), dt)
        capi.check(self._lib.qcnn_tap(self._h, which, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def layer_params(self, layer: int):
        b, m, s = C.c_int(), C.c_int(), C.c_int()
        capi.check(self._lib.qcnn_layer_params(self._h, layer, C.byref(b), C.byref(m), C.byref(s)))
        return b.value, m.value, s.value

    def set_exact(self, on) -> None:
        """From the next load on: 1 / True forces the float-pipeline kernels, 2 the clamped integer form of kernel A,
        0 / False returns to what the load-time proof chooses."""
        capi.check(self._lib.qcnn_set_exact(self._h, int(on)))

    @property
    def arith_paths(self):
        """(kernel A, kernel B): 0 integer epilogues, 1 float pipeline, 2 integer epilogues on clamped C2 accumulators."""
        v = capi.check(self._lib.qcnn_is_exact(self._h))
        return v & 15, v >> 4

    @property
    def is_exact(self) -> bool:
        """True when a fused kernel replays the float32 pipeline literally for the loaded model."""
        return 1 in self.arith_paths

    def bind_io(self, dev_y: int, dev_rec: int) -> None:
        capi.check(self._lib.qcnn_bind_io(self._h, C.c_void_p(dev_y), C.c_void_p(dev_rec)))

    def forward_blu_clip(self, frames: np.ndarray, ori: np.ndarray = None):
        """load_data + forward_blu + download for every frame of a host clip [n][H][W] (uint8), pipelined.
        Returns rec, or (rec, sse_before, sse_after) when the original clip is given."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n = frames.shape[0]
        assert frames.shape[1:] == (self.height, self.width)
        rec = np.empty_like(frames)
        hw = self.height * self.width
        if ori is None:
            capi.check(self._lib.qcnn_forward_blu_clip(self._h, frames.ctypes.data, rec.ctypes.data, n, hw, None, None, None))
            return rec
        ori = np.ascontiguousarray(ori, dtype=np.uint8)
        assert ori.shape == frames.shape
        before, after = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        capi.check(self._lib.qcnn_forward_blu_clip(self._h, frames.ctypes.data, rec.ctypes.data, n, hw, ori.ctypes.data,
                                                   before.ctypes.data, after.ctypes.data))
        return rec, before, after

    def batch_group(self, set_to: int = 0) -> int:
        """Frames per kernel launch of forward_blu_batch; set_to > 0 changes it."""
        cur = C.c_int()
        capi.check(self._lib.qcnn_batch_group(self._h, int(set_to), C.byref(cur)))
        return cur.value

    def set_timing(self, on: bool) -> None:
        capi.check(self._lib.qcnn_set_timing(self._h, int(on)))

    def kernel_times_us(self, reset: bool = True):
        a, b, n = C.c_double(), C.c_double(), C.c_int()
        capi.check(self._lib.qcnn_kernel_times_us(self._h, C.byref(a), C.byref(b), C.byref(n), int(reset)))
        return a.value, b.value, n.value

    def sse_device(self, dev_a: int, dev_b: int, n: int) -> int:
        s = C.c_uint64()
        capi.check(self._lib.qcnn_sse_device(self._h, C.c_void_p(dev_a), C.c_void_p(dev_b), n, C.byref(s)))
        return s.value

    @property
    def x_rec_ptr(self) -> int:
        return self._lib.qcnn_recon_device_ptr(self._h)

    @property
    def x_ptr(self) -> int:
        return self._lib.qcnn_input_device_ptr(self._h)

    @property
    def stream(self) -> int:
        return self._lib.qcnn_stream(self._h)

    def close(self) -> None:
        if self._h:
            self._lib.qcnn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


__all__ = ["qvrcnn", "LAYERS"]
