"""qcnn_forward_blu_batch_noalloc (static device scratch, no allocation) against qcnn_forward_blu_batch.

Two calls in a row on one GPU, the second on a smaller input, each compared byte for byte with the
allocating entry point and, for the first frames, with the CPU oracle; plus the size limit."""
import ctypes as C

import numpy as np
import pytest

from qcnn_gpu_amd import capi, synth
from qcnn_gpu_amd.qvrcnn import qvrcnn

pytestmark = pytest.mark.gpu

NOALLOC_SCRATCH_BYTES = 96 << 20      # QCNN_NOALLOC_SCRATCH_BYTES of include/qcnn.h


def _run_both(lib, net, frames):
    """frames: uint8 [n][h][w] -> (rec of the allocating entry point, rec of the static-scratch one)"""
    n, h, w = frames.shape
    hw = h * w
    d_in, d_a, d_b = C.c_void_p(), C.c_void_p(), C.c_void_p()
    for d in (d_in, d_a, d_b):
        capi.check(lib.qcnn_dev_alloc(C.byref(d), n * hw))
    try:
        capi.check(lib.qcnn_dev_h2d(d_in, np.ascontiguousarray(frames).ctypes.data, n * hw))
        capi.check(lib.qcnn_dev_memset(d_a, 0x11, n * hw))
        capi.check(lib.qcnn_dev_memset(d_b, 0x22, n * hw))
        capi.check(lib.qcnn_forward_blu_batch(net._h, d_in, d_a, n, hw))
        capi.check(lib.qcnn_sync(net._h))
        capi.check(lib.qcnn_forward_blu_batch_noalloc(net._h, d_in, d_b, n, hw))
        capi.check(lib.qcnn_sync(net._h))
        ra = np.empty((n, h, w), np.uint8)
        rb = np.empty((n, h, w), np.uint8)
        capi.check(lib.qcnn_dev_d2h(ra.ctypes.data, d_a, n * hw))
        capi.check(lib.qcnn_dev_d2h(rb.ctypes.data, d_b, n * hw))
        return ra, rb
    finally:
        for d in (d_in, d_a, d_b):
            capi.check(lib.qcnn_dev_free(d))


def test_noalloc_matches_batch_on_two_calls_in_a_row():
    from oracle import c_oracle
    lib = capi.load()
    qp = 37
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    # first call: 5 frames of 416x240 (the scratch holds them all: one launch per kernel)
    h1, w1 = 240, 416
    _, anc1 = synth.synth_frames(h1, w1, qp, frames=5)
    with qvrcnn(0, 1, 1, h1, w1) as net:
        net.load_static_para(mb)
        ra, rb = _run_both(lib, net, anc1)
        assert np.array_equal(ra, rb)
        assert np.array_equal(rb[0], c_oracle.forward_blu(mb, anc1[0]))
        assert np.array_equal(rb[4], c_oracle.forward_blu(mb, anc1[4]))
    # second call, smaller input (fewer and smaller frames, another engine on the same GPU: same array,
    # which still holds the first call's conc1)
    h2, w2 = 67, 101
    _, anc2 = synth.synth_frames(h2, w2, qp, frames=2, seed=7)
    with qvrcnn(0, 1, 1, h2, w2) as net:
        net.load_static_para(mb)
        ra, rb = _run_both(lib, net, anc2)
        assert np.array_equal(ra, rb)
        assert np.array_equal(rb[1], c_oracle.forward_blu(mb, anc2[1]))


def test_noalloc_groups_frames_by_scratch_capacity():
    # 1280x720: one frame needs 44.2 MB of conc1, the scratch holds two; five frames = three launches per kernel
    lib = capi.load()
    qp = 27
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    h, w = 720, 1280
    assert 2 * 48 * h * w <= NOALLOC_SCRATCH_BYTES < 3 * 48 * h * w
    _, anc = synth.synth_frames(h, w, qp, frames=5)
    with qvrcnn(0, 1, 1, h, w) as net:
        net.load_static_para(mb)
        ra, rb = _run_both(lib, net, anc)
        assert np.array_equal(ra, rb)


def test_noalloc_rejects_frames_beyond_the_scratch():
    lib = capi.load()
    mb = synth.synth_model(32).to_nchw_vect_c_bytes()
    h, w = 1088, 1936                      # 48*h*w > 96 MiB
    assert 48 * h * w > NOALLOC_SCRATCH_BYTES
    with qvrcnn(0, 1, 1, h, w) as net:
        net.load_static_para(mb)
        d = C.c_void_p()
        capi.check(lib.qcnn_dev_alloc(C.byref(d), 2 * h * w))
        try:
            rc = lib.qcnn_forward_blu_batch_noalloc(net._h, d, C.c_void_p(d.value + h * w), 1, h * w)
            assert rc == -6                # QCNN_E_UNSUPPORTED
            assert b"static scratch" in lib.qcnn_last_error()
        finally:
            capi.check(lib.qcnn_dev_free(d))
