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


def test_noalloc_allocates_nothing_and_creates_no_events():
    """The property that names the entry point: with the kernel timing of the engine switched on (the path that would
    create events in the allocating entry point), a call leaves the free device memory unchanged and records no
    timed pass; the allocating entry point on the same engine does record one."""
    lib = capi.load()
    qp = 32
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    h, w, n = 240, 416, 6
    _, anc = synth.synth_frames(h, w, qp, frames=n)
    hw = h * w
    with qvrcnn(0, 1, 1, h, w) as net:
        net.load_static_para(mb)
        d_in, d_out = C.c_void_p(), C.c_void_p()
        capi.check(lib.qcnn_dev_alloc(C.byref(d_in), n * hw))
        capi.check(lib.qcnn_dev_alloc(C.byref(d_out), n * hw))
        try:
            capi.check(lib.qcnn_dev_h2d(d_in, np.ascontiguousarray(anc).ctypes.data, n * hw))
            # first call: whatever the runtime sets up once per kernel (code objects, queues) happens here
            capi.check(lib.qcnn_forward_blu_batch_noalloc(net._h, d_in, d_out, n, hw))
            capi.check(lib.qcnn_sync(net._h))
            net.set_timing(True)
            net.kernel_times_us(reset=True)
            free0, free1 = C.c_size_t(), C.c_size_t()
            capi.check(lib.qcnn_dev_mem_info(C.byref(free0), None))
            for _ in range(3):
                capi.check(lib.qcnn_forward_blu_batch_noalloc(net._h, d_in, d_out, n, hw))
            capi.check(lib.qcnn_sync(net._h))
            capi.check(lib.qcnn_dev_mem_info(C.byref(free1), None))
            assert free1.value == free0.value, (free0.value, free1.value)
            assert net.kernel_times_us()[2] == 0          # no timing event was created or recorded
            capi.check(lib.qcnn_forward_blu_batch(net._h, d_in, d_out, n, hw))
            assert net.kernel_times_us(reset=True)[2] >= 1
            net.set_timing(False)
            # the fused conc1 tap reads the image the LAST pass wrote: the engine's own after the allocating
            # entry point, the static scratch after this one (same frame 0, same bytes)
            t_alloc = net.tap(capi.TAP_FUSED_CONC1)
            capi.check(lib.qcnn_forward_blu_batch_noalloc(net._h, d_in, d_out, 1, hw))
            assert np.array_equal(net.tap(capi.TAP_FUSED_CONC1), t_alloc)
        finally:
            capi.check(lib.qcnn_dev_free(d_in))
            capi.check(lib.qcnn_dev_free(d_out))


def test_host_register_round_trip():
    """qcnn_host_register / qcnn_host_unregister on memory the caller owns (a numpy clip): the pipelined clip entry
    point gives the same bytes on the registered buffers as on pageable ones."""
    lib = capi.load()
    qp = 27
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    h, w, n = 96, 160, 7
    _, anc = synth.synth_frames(h, w, qp, frames=n)
    clip = np.ascontiguousarray(anc)
    rec_pageable = np.empty_like(clip)
    rec_pinned = np.empty_like(clip)
    with qvrcnn(0, 1, 1, h, w) as net:
        net.load_static_para(mb)
        net.batch_group(3)
        # registered first: the buffers are fresh, no copy has touched them yet
        rc = lib.qcnn_host_register(clip.ctypes.data, clip.nbytes)
        if rc != 0:
            # Some machines refuse to page-lock user memory at all (tools/register_probe.py: every buffer, aligned or
            # not, small or large, gets "invalid argument" where the driver has no user-pointer support).  The entry
            # point must then fail cleanly, and the pageable path must still give the right bytes.
            assert rc == -4 and b"hipHostRegister" in lib.qcnn_last_error()
            capi.check(lib.qcnn_forward_blu_clip(net._h, clip.ctypes.data, rec_pageable.ctypes.data, n, h * w, None, None, None))
            from oracle import c_oracle
            assert np.array_equal(rec_pageable[n - 1], c_oracle.forward_blu(mb, clip[n - 1]))
            pytest.skip("hipHostRegister is refused on this machine: " + lib.qcnn_last_error().decode(errors="replace"))
        capi.check(lib.qcnn_host_register(rec_pinned.ctypes.data, rec_pinned.nbytes))
        try:
            capi.check(lib.qcnn_forward_blu_clip(net._h, clip.ctypes.data, rec_pinned.ctypes.data, n, h * w, None, None, None))
        finally:
            capi.check(lib.qcnn_host_unregister(clip.ctypes.data))
            capi.check(lib.qcnn_host_unregister(rec_pinned.ctypes.data))
        capi.check(lib.qcnn_forward_blu_clip(net._h, clip.ctypes.data, rec_pageable.ctypes.data, n, h * w, None, None, None))
        assert np.array_equal(rec_pinned, rec_pageable)
        from oracle import c_oracle
        assert np.array_equal(rec_pinned[n - 1], c_oracle.forward_blu(mb, clip[n - 1]))
    assert lib.qcnn_host_unregister(clip.ctypes.data) != 0   # no longer registered: an error code, not a crash


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
