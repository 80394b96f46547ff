"""ctypes binding of the C ABI in include/qcnn.h (qcnn_gpu_amd/lib/libqcnn.so).

There is no CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libqcnn.so")

# every symbol include/qcnn.h declares
SYMBOLS = (
    "qcnn_last_error", "qcnn_version", "qcnn_device_count", "qcnn_device_pci_bus_id", "qcnn_create", "qcnn_destroy", "qcnn_load_static",
    "qcnn_load_static_mem", "qcnn_upload", "qcnn_upload_async", "qcnn_forward_blu", "qcnn_forward_blu_batch",
    "qcnn_forward_blu_batch_noalloc", "qcnn_forward_blu_layered", "qcnn_batch_group", "qcnn_forward_blu_clip", "qcnn_clip_reserve",
    "qcnn_sync", "qcnn_recon_device_ptr", "qcnn_input_device_ptr", "qcnn_bind_io", "qcnn_download",
    "qcnn_download_async", "qcnn_sse_device", "qcnn_tap_bytes", "qcnn_tap", "qcnn_tap_device_ptr",
    "qcnn_layer_params", "qcnn_layer_w_device_ptr", "qcnn_layer_b_device_ptr", "qcnn_set_exact", "qcnn_is_exact",
    "qcnn_stream", "qcnn_set_timing", "qcnn_kernel_times_us", "qcnn_op_ppro", "qcnn_op_conv_static",
    "qcnn_op_quant_blu", "qcnn_op_apply_res_y", "qcnn_set_device", "qcnn_dev_alloc", "qcnn_dev_free", "qcnn_dev_h2d",
    "qcnn_dev_d2h", "qcnn_dev_memset", "qcnn_dev_sync", "qcnn_dev_mem_info", "qcnn_host_alloc_pinned", "qcnn_host_free_pinned",
    "qcnn_host_register", "qcnn_host_unregister",
    "qcnn_hwcn2nchw_vect_c", "qcnn_model_qfp_hwcn2nchw_vect_c",
    # dynamic-quantisation path
    "qcnn_load_para", "qcnn_load_para_mem", "qcnn_forward", "qcnn_forward_layered", "qcnn_dynamic_state", "qcnn_op_activate",
    "qcnn_op_find_max", "qcnn_op_quant_div", "qcnn_op_mul_shift", "qcnn_op_adjust_basic", "qcnn_op_adjust_output",
    "qcnn_op_adjust_output_static", "qcnn_op_apply_res", "qcnn_concat_steps", "qcnn_layer_b_adj_device_ptr",
    # float32 configuration
    "qcnn_load_float", "qcnn_load_float_mem", "qcnn_forward_float", "qcnn_float_w_device_ptr", "qcnn_float_b_device_ptr",
    "qcnn_float_u_device_ptr", "qcnn_op_ppro_float", "qcnn_op_conv_float", "qcnn_op_apply_res_float", "qcnn_hwcn2nchw",
    "qcnn_model_hwcn2nchw",
)

TAP_X_PPRO, TAP_U_C1, TAP_U_C2_1, TAP_U_C2_2, TAP_U_C3_1, TAP_U_C3_2, TAP_U_C4 = range(7)
TAP_V_C1, TAP_CONC1, TAP_CONC2, TAP_X_REC, TAP_FUSED_CONC1, TAP_V_C4 = 7, 8, 9, 10, 11, 12
TAP_F_X_PPRO, TAP_F_U_C1, TAP_F_CONC1, TAP_F_CONC2, TAP_F_U_C4 = 13, 14, 15, 16, 17


class DynamicState(C.Structure):
    """qcnn_dynamic_state_t of include/qcnn.h."""
    _fields_ = [("stepw", C.c_int32 * 4), ("stepy", C.c_int32 * 4), ("step_w", C.c_int32 * 6),
                ("step_y", C.c_int32 * 6), ("max_u", C.c_int32 * 6), ("b_adj", C.c_float * 161)]


class QcnnError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"qcnn error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """Loads libqcnn.so and declares the signatures.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} is missing: build it with `make -C qcnn_gpu_amd/csrc` "
                                "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
    L.qcnn_last_error.restype = C.c_char_p
    L.qcnn_version.restype = C.c_char_p
    L.qcnn_device_count.argtypes = [C.POINTER(i)]
    L.qcnn_device_pci_bus_id.argtypes = [i, C.c_char_p, i]
    L.qcnn_create.argtypes = [i, i, i, i, i, C.POINTER(vp)]
    L.qcnn_destroy.argtypes = [vp]
    L.qcnn_load_static.argtypes = [vp, C.c_char_p]
    L.qcnn_load_static_mem.argtypes = [vp, vp, sz]
    L.qcnn_upload.argtypes = [vp, vp]
    L.qcnn_upload_async.argtypes = [vp, vp]
    L.qcnn_forward_blu.argtypes = [vp]
    L.qcnn_forward_blu_layered.argtypes = [vp]
    L.qcnn_forward_blu_batch.argtypes = [vp, vp, vp, i, sz]
    L.qcnn_forward_blu_batch_noalloc.argtypes = [vp, vp, vp, i, sz]
    L.qcnn_batch_group.argtypes = [vp, i, C.POINTER(i)]
    L.qcnn_forward_blu_clip.argtypes = [vp, vp, vp, i, sz, vp, vp, vp]
    L.qcnn_clip_reserve.argtypes = [vp, i, i]
    L.qcnn_sync.argtypes = [vp]
    L.qcnn_recon_device_ptr.restype = vp
    L.qcnn_recon_device_ptr.argtypes = [vp]
    L.qcnn_input_device_ptr.restype = vp
    L.qcnn_input_device_ptr.argtypes = [vp]
    L.qcnn_bind_io.argtypes = [vp, vp, vp]
    L.qcnn_download.argtypes = [vp, vp]
    L.qcnn_download_async.argtypes = [vp, vp]
    L.qcnn_sse_device.argtypes = [vp, vp, vp, sz, C.POINTER(C.c_uint64)]
    L.qcnn_tap_bytes.argtypes = [vp, i, C.POINTER(sz)]
    L.qcnn_tap.argtypes = [vp, i, vp, sz]
    L.qcnn_tap_device_ptr.restype = vp
    L.qcnn_tap_device_ptr.argtypes = [vp, i]
    L.qcnn_layer_params.argtypes = [vp, i, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.qcnn_layer_w_device_ptr.restype = vp
    L.qcnn_layer_w_device_ptr.argtypes = [vp, i]
    L.qcnn_layer_b_device_ptr.restype = vp
    L.qcnn_layer_b_device_ptr.argtypes = [vp, i]
    L.qcnn_set_exact.argtypes = [vp, i]
    L.qcnn_is_exact.argtypes = [vp]
    L.qcnn_stream.restype = vp
    L.qcnn_stream.argtypes = [vp]
    L.qcnn_set_timing.argtypes = [vp, i]
    L.qcnn_kernel_times_us.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i), i]
    L.qcnn_op_ppro.argtypes = [vp, vp, sz, vp]
    L.qcnn_op_conv_static.argtypes = [vp, vp, vp, vp, i, i, i, i, i, vp]
    L.qcnn_op_quant_blu.argtypes = [vp, vp, sz, i, i, i, i, vp]
    L.qcnn_op_apply_res_y.argtypes = [vp, vp, vp, i, i, sz, vp]
    L.qcnn_set_device.argtypes = [i]
    L.qcnn_dev_alloc.argtypes = [C.POINTER(vp), sz]
    L.qcnn_dev_free.argtypes = [vp]
    L.qcnn_dev_h2d.argtypes = [vp, vp, sz]
    L.qcnn_dev_d2h.argtypes = [vp, vp, sz]
    L.qcnn_dev_memset.argtypes = [vp, i, sz]
    L.qcnn_dev_mem_info.argtypes = [C.POINTER(sz), C.POINTER(sz)]
    L.qcnn_host_alloc_pinned.argtypes = [C.POINTER(vp), sz]
    L.qcnn_host_free_pinned.argtypes = [vp]
    L.qcnn_host_register.argtypes = [vp, sz]
    L.qcnn_host_unregister.argtypes = [vp]
    L.qcnn_hwcn2nchw_vect_c.restype = C.c_long
    L.qcnn_hwcn2nchw_vect_c.argtypes = [vp, i, i, i, i, vp]
    L.qcnn_model_qfp_hwcn2nchw_vect_c.argtypes = [C.c_char_p, C.c_char_p, i]
    L.qcnn_load_para.argtypes = [vp, C.c_char_p]
    L.qcnn_load_para_mem.argtypes = [vp, vp, sz]
    L.qcnn_forward.argtypes = [vp]
    L.qcnn_forward_layered.argtypes = [vp]
    L.qcnn_dynamic_state.argtypes = [vp, C.POINTER(DynamicState)]
    L.qcnn_op_activate.argtypes = [vp, sz, vp]
    L.qcnn_op_find_max.argtypes = [vp, vp, sz, C.POINTER(C.c_float), vp]
    L.qcnn_op_quant_div.argtypes = [vp, vp, sz, i, i, vp]
    L.qcnn_op_mul_shift.argtypes = [vp, vp, sz, i, i, vp]
    L.qcnn_op_adjust_basic.argtypes = [vp, vp, vp, vp, i, i, vp]
    L.qcnn_op_adjust_output.argtypes = [vp, vp, vp, vp, i, sz, vp]
    L.qcnn_op_adjust_output_static.argtypes = [vp, vp, i, i, sz, vp]
    L.qcnn_op_apply_res.argtypes = [vp, vp, vp, sz, vp]
    L.qcnn_layer_b_adj_device_ptr.restype = vp
    L.qcnn_layer_b_adj_device_ptr.argtypes = [vp, i]
    L.qcnn_concat_steps.argtypes = [C.c_float, C.c_float, i, i, C.POINTER(i), C.POINTER(i)]
    L.qcnn_load_float.argtypes = [vp, C.c_char_p]
    L.qcnn_load_float_mem.argtypes = [vp, vp, sz]
    L.qcnn_forward_float.argtypes = [vp]
    for fn in (L.qcnn_float_w_device_ptr, L.qcnn_float_b_device_ptr, L.qcnn_float_u_device_ptr):
        fn.restype = vp
        fn.argtypes = [vp, i]
    L.qcnn_op_ppro_float.argtypes = [vp, vp, sz, vp]
    L.qcnn_op_conv_float.argtypes = [vp, vp, vp, vp, i, i, i, i, i, i, vp]
    L.qcnn_op_apply_res_float.argtypes = [vp, vp, vp, sz, vp]
    L.qcnn_hwcn2nchw.restype = C.c_long
    L.qcnn_hwcn2nchw.argtypes = [vp, i, i, i, i, vp]
    L.qcnn_model_hwcn2nchw.argtypes = [C.c_char_p, C.c_char_p, i]
    _lib = L
    return L


def check(rc: int) -> int:
    if rc < 0:
        raise QcnnError(rc, load().qcnn_last_error().decode(errors="replace"))
    return rc


def pci_bus_id(gpu: int) -> str:
    """PCI bus id of device `gpu` (qcnn_device_pci_bus_id)."""
    buf = C.create_string_buffer(64)
    check(load().qcnn_device_pci_bus_id(int(gpu), buf, 64))
    return buf.value.decode()
