"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle -- bit-exact.

Tolerance: none.  Every comparison below is np.array_equal on bytes (uint8 / int8) or on float32
bit patterns of integer-valued accumulators.
"""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

from oracle import c_oracle
from qcnn_gpu_amd import capi, formats, synth
from qcnn_gpu_amd.qvrcnn import qvrcnn

pytestmark = pytest.mark.gpu

U_TAPS = [capi.TAP_U_C1, capi.TAP_U_C2_1, capi.TAP_U_C2_2, capi.TAP_U_C3_1, capi.TAP_U_C3_2, capi.TAP_U_C4]


def _frame(h, w, qp, seed=1):
    return synth.synth_frames(h, w, qp, seed=seed)[1][0]


def _check_layered(net, mb, y):
    rec_o, taps_o = c_oracle.forward_blu(mb, y, True)
    net.load_data(y)
    net.forward_blu_layered()
    net.synchronize()
    assert np.array_equal(net.tap(capi.TAP_X_PPRO), taps_o["x_ppro"])
    for l, t in enumerate(U_TAPS):
        got = net.tap(t)
        assert np.array_equal(got.view(np.uint32), taps_o["u"][l].view(np.uint32)), f"u[{l}]"
    assert np.array_equal(net.tap(capi.TAP_V_C1), taps_o["v1"])
    assert np.array_equal(net.tap(capi.TAP_CONC1), taps_o["conc1"])
    assert np.array_equal(net.tap(capi.TAP_CONC2), taps_o["conc2"])
    assert np.array_equal(net.download(), rec_o)
    return rec_o, taps_o


def _check_fused(net, y, rec_o, conc1_o):
    net.load_data(y)
    net.forward_blu()
    net.synchronize()
    conc1 = net.tap(capi.TAP_FUSED_CONC1)
    bad = np.argwhere(conc1 != conc1_o)
    assert bad.size == 0, f"fused conc1 differs at {bad[:5].tolist()} ({len(bad)} bytes)"
    rec = net.download()
    bad = np.argwhere(rec != rec_o)
    assert bad.size == 0, f"fused x_rec differs at {bad[:5].tolist()} ({len(bad)} pixels)"


def test_mfma_device_present():
    n = C.c_int()
    capi.check(capi.load().qcnn_device_count(C.byref(n)))
    assert n.value >= 1


@pytest.mark.parametrize("qp", [22, 27, 32, 37])
def test_golden_case_layered_and_fused(qp, golden_dir):
    case = np.load(f"{golden_dir}/synth_case_qp{qp}.npz")
    with open(f"{golden_dir}/synth_model_qp{qp}.data", "rb") as f:
        mb = f.read()
    y = case["anchor"]
    with qvrcnn(0, 1, 1, *y.shape) as net:
        net.load_static_para(f"{golden_dir}/synth_model_qp{qp}.data")
        assert net.layer_params(0) == synth.QUANT_TRIPLES[qp][0]
        rec_o, taps_o = _check_layered(net, mb, y)
        assert np.array_equal(rec_o, case["rec"])
        crcs = dict(zip(case["crc_names"].tolist(), case["crc_values"].tolist()))
        assert zlib.crc32(net.tap(capi.TAP_CONC2).tobytes()) == crcs["conc2"]
        assert not net.is_exact                       # realistic weights: integer epilogues proved safe
        _check_fused(net, y, case["rec"], case["conc1"])


@pytest.mark.parametrize("shape", [(1, 1), (2, 3), (5, 7), (16, 32), (18, 30), (17, 31), (19, 33), (20, 32), (21, 31), (36, 60),
                                   (37, 61), (40, 64), (41, 65), (59, 97), (67, 131), (240, 416)])
def test_fused_matches_oracle_on_ragged_sizes(shape):
    qp = 37
    m = synth.synth_model(qp)
    mb = m.to_nchw_vect_c_bytes()
    y = _frame(shape[0], shape[1], qp, seed=shape[0] * 7 + shape[1])
    rec_o, taps_o = c_oracle.forward_blu(mb, y, True)
    with qvrcnn(0, 1, 1, *shape) as net:
        net.load_static_para(mb)
        _check_fused(net, y, rec_o, taps_o["conc1"])
        if shape[0] * shape[1] <= 40 * 60:
            _check_layered(net, mb, y)


@pytest.mark.parametrize("qp", [22, 27, 32])
def test_fused_float_pipeline_variant(qp):
    """EXACT kernels (literal float32 accumulator pipeline) give the same bytes as the integer ones."""
    m = synth.synth_model(qp)
    mb = m.to_nchw_vect_c_bytes()
    y = _frame(45, 77, qp)
    rec_o, taps_o = c_oracle.forward_blu(mb, y, True)
    with qvrcnn(0, 1, 1, 45, 77) as net:
        net.set_exact(True)
        net.load_static_para(mb)
        assert net.is_exact
        _check_fused(net, y, rec_o, taps_o["conc1"])
        net.set_exact(2)          # kernel A with its C2 accumulators clamped first
        net.load_static_para(mb)
        assert net.arith_paths == (2, 0)
        _check_fused(net, y, rec_o, taps_o["conc1"])
        net.set_exact(False)
        net.load_static_para(mb)
        assert not net.is_exact and net.arith_paths == (0, 0)
        _check_fused(net, y, rec_o, taps_o["conc1"])


@pytest.mark.parametrize("kind", ["maxmag", "identity", "wide", "balanced", "heavy", "c4big"])
def test_adversarial_models(kind):
    m = synth.adversarial_model(kind)
    mb = m.to_nchw_vect_c_bytes()
    if kind == "balanced":
        # sum|w| * 127 alone passes 2^23 (C2_1, C2_2), the signed sums stay below it in every layer
        mags = [np.abs(lp.w.astype(np.int64)).reshape(lp.w.shape[0], -1).sum(1).max() * 127 for lp in m.layers]
        pos = [np.maximum(lp.w.astype(np.int64), 0).reshape(lp.w.shape[0], -1).sum(1).max() * 127 for lp in m.layers]
        neg = [np.maximum(-lp.w.astype(np.int64), 0).reshape(lp.w.shape[0], -1).sum(1).max() * 127 for lp in m.layers]
        assert mags[1] >= 1 << 23 and mags[2] >= 1 << 23
        assert max(pos[1:] + neg[1:]) + 8000 + 70 < 1 << 23
    frames = [np.zeros((21, 35), np.uint8), np.full((21, 35), 255, np.uint8), _frame(21, 35, 37)]
    with qvrcnn(0, 1, 1, 21, 35) as net:
        net.load_static_para(mb)
        if kind == "maxmag":
            assert net.is_exact               # accumulators can pass 2^24: float pipeline is selected
        if kind in ("wide", "balanced"):
            assert net.arith_paths == (0, 0)  # inside the 24-bit operand range: integer requantiser
        if kind == "c4big":
            assert net.arith_paths == (0, 1)  # integers in kernel A, the literal (wrapping) residual arithmetic in kernel B
        if kind == "heavy":
            # C2 accumulator bounds beyond 2^24, everything else in range: clamped integers in kernel A, plain in B
            pos = [np.maximum(lp.w.astype(np.int64), 0).reshape(lp.w.shape[0], -1).sum(1).max() * 127 for lp in m.layers]
            assert pos[2] > 1 << 24 and net.arith_paths == (2, 0)
        for y in frames:
            rec_o, taps_o = _check_layered(net, mb, y)
            _check_fused(net, y, rec_o, taps_o["conc1"])


def test_all_four_qp_models_back_to_back():
    """configs[2]: the four QP models loaded one after another into one engine."""
    h, w = 72, 100
    with qvrcnn(0, 1, 1, h, w) as net:
        for qp in (22, 27, 32, 37, 22):
            mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
            y = _frame(h, w, qp)
            rec_o, taps_o = c_oracle.forward_blu(mb, y, True)
            net.load_static_para(mb)
            _check_fused(net, y, rec_o, taps_o["conc1"])


def test_full_hd_frame_bit_exact():
    """configs[1]: one 1920x1080 Y frame, QP37, memcmp-equal with the oracle, + PSNR equality."""
    qp, h, w = 37, 1080, 1920
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    ori, anc = synth.synth_frames(h, w, qp)
    rec_o, taps_o = c_oracle.forward_blu(mb, anc[0], True)
    with qvrcnn(0, 1, 1, h, w) as net:
        net.load_static_para(mb)
        _check_fused(net, anc[0], rec_o, taps_o["conc1"])
        # idempotence of the engine state: a second pass gives the same bytes
        net.forward_blu()
        net.synchronize()
        rec = net.download()
        assert np.array_equal(rec, rec_o)
        # PSNR "to the last printed decimal": identical integer SSE => identical double
        lib = capi.load()
        d_ori = C.c_void_p()
        capi.check(lib.qcnn_dev_alloc(C.byref(d_ori), h * w))
        capi.check(lib.qcnn_dev_h2d(d_ori, ori[0].ctypes.data, h * w))
        sse = net.sse_device(net.x_rec_ptr, d_ori.value, h * w)
        capi.check(lib.qcnn_dev_free(d_ori))
        assert sse == c_oracle.sse(rec_o, ori[0])
        assert formats.psnr_from_sse(sse, h * w) == c_oracle.psnr(rec_o, ori[0])
        # layered mode at full size agrees too
        net.forward_blu_layered()
        net.synchronize()
        assert np.array_equal(net.download(), rec_o)
        assert np.array_equal(net.tap(capi.TAP_CONC2), taps_o["conc2"])
        # the clamped integer form of kernel A and the float pipeline on the same frame
        for mode in (2, 1):
            net.set_exact(mode)
            net.load_static_para(mb)
            assert net.arith_paths == ((2, 0) if mode == 2 else (1, 1))
            _check_fused(net, anc[0], rec_o, taps_o["conc1"])


def test_full_hd_heavy_model_bit_exact():
    """A 1920x1080 frame through the model whose C2 accumulator bounds pass 2^24 (clamped integer form of kernel A
    chosen by the load-time proof), white and textured input, memcmp-equal with the oracle."""
    h, w = 1080, 1920
    mb = synth.adversarial_model("heavy").to_nchw_vect_c_bytes()
    y = np.array(_frame(h, w, 37), np.uint8, order="C")
    y[: h // 3] = 255          # a white band: every C1 channel with positive weights saturates, C2_2 passes 2^24
    rec_o, taps_o = c_oracle.forward_blu(mb, y, True)
    with qvrcnn(0, 1, 1, h, w) as net:
        net.load_static_para(mb)
        assert net.arith_paths == (2, 0)
        _check_fused(net, y, rec_o, taps_o["conc1"])


def test_4k_frame_qp32_bit_exact():
    """configs[3] on one GPU: a 3840x2160 Y frame, QP32, memcmp-equal with the oracle; a two-frame batch
    through one launch gives the same bytes for both frames."""
    qp, h, w = 32, 2160, 3840
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    anc = synth.synth_frames(h, w, qp, frames=1, seed=4)[1]
    rec_o = c_oracle.forward_blu(mb, np.ascontiguousarray(anc[0]))
    lib = capi.load()
    with qvrcnn(0, 1, 1, h, w) as net:
        net.load_static_para(mb)
        net.load_data(np.ascontiguousarray(anc[0]))
        net.forward_blu()
        rec = net.download()
        bad = np.argwhere(rec != rec_o)
        assert bad.size == 0, f"4K x_rec differs at {bad[:5].tolist()} ({len(bad)} pixels)"
        d_in, d_out = C.c_void_p(), C.c_void_p()
        capi.check(lib.qcnn_dev_alloc(C.byref(d_in), 2 * h * w))
        capi.check(lib.qcnn_dev_alloc(C.byref(d_out), 2 * h * w))
        for i in range(2):
            capi.check(lib.qcnn_dev_h2d(C.c_void_p(d_in.value + i * h * w), anc[0].ctypes.data, h * w))
        net.forward_blu_batch(d_in.value, d_out.value, 2, h * w)
        net.synchronize()
        out = np.empty((2, h, w), np.uint8)
        capi.check(lib.qcnn_dev_d2h(out.ctypes.data, d_out, 2 * h * w))
        assert np.array_equal(out[0], rec_o) and np.array_equal(out[1], rec_o)
        capi.check(lib.qcnn_dev_free(d_in))
        capi.check(lib.qcnn_dev_free(d_out))


def test_8k_frame_windows_equal_their_crops():
    """The largest frame the engine takes to within a row (7680x4320: conc1 offsets up to 1.6e9, under the 2^31 the
    buffer descriptors allow) through the size-independent property that every output pixel depends on a 13x13
    neighbourhood only (5x5, 5x5, 3x3, 3x3): windows of the 8K result equal the oracle's result on the window
    grown by 6 pixels (cut at the frame edges, where the zero padding is the frame's own)."""
    qp, h, w = 37, 4320, 7680
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    rng = np.random.default_rng(8192)
    frame = rng.integers(0, 256, (h, w), dtype=np.uint8)
    with qvrcnn(0, 1, 1, h, w) as net:
        net.load_static_para(mb)
        net.load_data(frame)
        net.forward_blu()
        rec = net.download()
    m = 6
    for y0, x0, hh, ww in [(0, 0, 150, 200), (h - 170, w - 210, 170, 210), (2000, 3700, 128, 300), (h - 90, 0, 90, 120),
                           (0, w - 100, 75, 100), (4000, 7000, 64, 64)]:
        ya, yb, xa, xb = max(y0 - m, 0), min(y0 + hh + m, h), max(x0 - m, 0), min(x0 + ww + m, w)
        crop = np.ascontiguousarray(frame[ya:yb, xa:xb])
        ref = c_oracle.forward_blu(mb, crop)
        # interior of the crop: everything at least 6 pixels from a cut that is not a frame edge
        ia = 0 if ya == 0 else m
        ib = (yb - ya) if yb == h else (yb - ya) - m
        ja = 0 if xa == 0 else m
        jb = (xb - xa) if xb == w else (xb - xa) - m
        got = rec[ya + ia:ya + ib, xa + ja:xa + jb]
        assert np.array_equal(got, ref[ia:ib, ja:jb]), f"window at ({y0}, {x0})"


@pytest.mark.parametrize("group", [1, 3, 16])
def test_batch_of_frames_any_group(group):
    """The frame loop of inference/kernel.cu:91-97 on resident frames: 7 distinct frames with a
    padded stride, grouped 1, 3 (ragged last group) and 16 (> batch) frames per kernel launch."""
    qp, h, w, nf = 27, 53, 95, 7
    stride = h * w + 37
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    frames = synth.synth_frames(h, w, qp, frames=nf, seed=99)[1]
    lib = capi.load()
    d_in, d_out = C.c_void_p(), C.c_void_p()
    capi.check(lib.qcnn_dev_alloc(C.byref(d_in), nf * stride))
    capi.check(lib.qcnn_dev_alloc(C.byref(d_out), nf * stride))
    capi.check(lib.qcnn_dev_memset(d_out, 0xAB, nf * stride))
    for i in range(nf):
        f = np.ascontiguousarray(frames[i])
        capi.check(lib.qcnn_dev_h2d(C.c_void_p(d_in.value + i * stride), f.ctypes.data, h * w))
    with qvrcnn(0, 1, 1, h, w) as net:
        net.load_static_para(mb)
        assert net.batch_group(group) == group
        net.forward_blu_batch(d_in.value, d_out.value, nf, stride)
        net.synchronize()
        out = np.empty(nf * stride, np.uint8)
        capi.check(lib.qcnn_dev_d2h(out.ctypes.data, d_out, nf * stride))
        for i in range(nf):
            rec_o = c_oracle.forward_blu(mb, np.ascontiguousarray(frames[i]))
            assert np.array_equal(out[i * stride:i * stride + h * w].reshape(h, w), rec_o), f"frame {i}"
            if i + 1 < nf:
                assert (out[i * stride + h * w:(i + 1) * stride] == 0xAB).all()   # padding untouched
        # a single-frame pass afterwards still works on the engine's own buffers
        net.load_data(np.ascontiguousarray(frames[2]))
        net.forward_blu()
        assert np.array_equal(net.download(), c_oracle.forward_blu(mb, np.ascontiguousarray(frames[2])))
    capi.check(lib.qcnn_dev_free(d_in))
    capi.check(lib.qcnn_dev_free(d_out))


@pytest.mark.parametrize("form", ["two_tiles", "one_tile", "tiled"])
def test_batch_group_above_sixteen_frames(form, monkeypatch):
    """A launch group above 16 frames (qcnn_batch_group / QCNN_BATCH_GROUP): kernel A walks all 37 distinct frames in
    one launch, kernel B the same frames in launches of 16, 16 and 5, in every form of kernel B."""
    if form == "tiled":
        monkeypatch.setenv("QCNN_B_TILED", "1")
    elif form == "one_tile":
        monkeypatch.setenv("QCNN_B_WIDE", "0")
    qp, h, w, nf = 32, 31, 47, 37
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    base = synth.synth_frames(h, w, qp, frames=8, seed=123)[1]
    frames = np.stack([np.roll(base[i % 8], (2 * i, 3 * i), (0, 1)) for i in range(nf)])
    lib = capi.load()
    d_in, d_out = C.c_void_p(), C.c_void_p()
    capi.check(lib.qcnn_dev_alloc(C.byref(d_in), nf * h * w))
    capi.check(lib.qcnn_dev_alloc(C.byref(d_out), nf * h * w))
    try:
        capi.check(lib.qcnn_dev_h2d(d_in, np.ascontiguousarray(frames).ctypes.data, nf * h * w))
        with qvrcnn(0, 1, 1, h, w) as net:
            net.load_static_para(mb)
            assert net.batch_group(40) == 40
            net.forward_blu_batch(d_in.value, d_out.value, nf, h * w)
            net.synchronize()
            out = np.empty((nf, h, w), np.uint8)
            capi.check(lib.qcnn_dev_d2h(out.ctypes.data, d_out, nf * h * w))
    finally:
        lib.qcnn_dev_free(d_in)
        lib.qcnn_dev_free(d_out)
    for i in range(nf):
        assert np.array_equal(out[i], c_oracle.forward_blu(mb, np.ascontiguousarray(frames[i]))), f"frame {i}"


def test_batch_calls_back_to_back():
    """Batch calls of several groups each, queued back to back on two pairs of buffers without a synchronisation
    in between: every call reuses the conc1 image of the one before it."""
    qp, h, w, nf = 32, 70, 123, 9
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    fa = synth.synth_frames(h, w, qp, frames=nf, seed=5)[1]
    fb = synth.synth_frames(h, w, qp, frames=nf, seed=6)[1]
    lib = capi.load()
    bufs = [C.c_void_p() for _ in range(4)]
    for b in bufs:
        capi.check(lib.qcnn_dev_alloc(C.byref(b), nf * h * w))
    for src, dst in ((fa, bufs[0]), (fb, bufs[1])):
        f = np.ascontiguousarray(src)
        capi.check(lib.qcnn_dev_h2d(dst, f.ctypes.data, nf * h * w))
    with qvrcnn(0, 1, 1, h, w) as net:
        net.load_static_para(mb)
        net.batch_group(2)
        for _ in range(3):
            net.forward_blu_batch(bufs[0].value, bufs[2].value, nf, h * w)
            net.forward_blu_batch(bufs[1].value, bufs[3].value, nf, h * w)
        net.synchronize()
        for src, dst in ((fa, bufs[2]), (fb, bufs[3])):
            out = np.empty((nf, h, w), np.uint8)
            capi.check(lib.qcnn_dev_d2h(out.ctypes.data, dst, nf * h * w))
            for i in range(nf):
                assert np.array_equal(out[i], c_oracle.forward_blu(mb, np.ascontiguousarray(src[i]))), f"frame {i}"
    for b in bufs:
        capi.check(lib.qcnn_dev_free(b))


@pytest.mark.parametrize("group", [1, 3, 16])
def test_host_clip_pipeline(group):
    """qcnn_forward_blu_clip: the H2D + forward_blu + D2H loop of inference/kernel.cu:89-101 as a pipeline,
    with the per-frame SSE sums of vrcnn_data::psnr_pf computed on the device."""
    qp, h, w, nf = 22, 41, 67, 8
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    ori, anc = synth.synth_frames(h, w, qp, frames=nf, seed=7)
    with qvrcnn(0, 1, 1, h, w) as net:
        net.load_static_para(mb)
        net.batch_group(group)
        rec, before, after = net.forward_blu_clip(anc, ori)
        rec2 = net.forward_blu_clip(anc)
    assert np.array_equal(rec, rec2)
    for i in range(nf):
        rec_o = c_oracle.forward_blu(mb, np.ascontiguousarray(anc[i]))
        assert np.array_equal(rec[i], rec_o), f"frame {i}"
        assert int(before[i]) == c_oracle.sse(np.ascontiguousarray(anc[i]), np.ascontiguousarray(ori[i]))
        assert int(after[i]) == c_oracle.sse(rec_o, np.ascontiguousarray(ori[i]))


@pytest.mark.parametrize("group,nf", [(16, 64), (16, 77), (8, 33)])
def test_host_clip_pipeline_ramped_chunks(group, nf):
    """A clip of at least four launch groups starts and ends with chunks of group/8, group/4 and group/2 frames
    (64 frames in groups of 16: 2 4 8 16 16 4 8 4 2 ...): every frame and both SSE sums against the oracle, and the
    same bytes with the ramp switched off."""
    qp, h, w = 32, 23, 45
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    ori_b, anc_b = synth.synth_frames(h, w, qp, frames=8, seed=11)
    ori = np.stack([np.roll(ori_b[i % 8], (3 * i, 5 * i), (0, 1)) for i in range(nf)])
    anc = np.stack([np.roll(anc_b[i % 8], (3 * i, 5 * i), (0, 1)) for i in range(nf)])
    with qvrcnn(0, 1, 1, h, w) as net:
        net.load_static_para(mb)
        net.batch_group(group)
        rec, before, after = net.forward_blu_clip(anc, ori)
        os.environ["QCNN_CLIP_NO_RAMP"] = "1"
        try:
            rec2 = net.forward_blu_clip(anc)
        finally:
            del os.environ["QCNN_CLIP_NO_RAMP"]
    assert np.array_equal(rec, rec2)
    for i in range(nf):
        rec_o = c_oracle.forward_blu(mb, np.ascontiguousarray(anc[i]))
        assert np.array_equal(rec[i], rec_o), f"frame {i}"
        assert int(before[i]) == c_oracle.sse(np.ascontiguousarray(anc[i]), np.ascontiguousarray(ori[i]))
        assert int(after[i]) == c_oracle.sse(rec_o, np.ascontiguousarray(ori[i]))


def test_random_shapes_and_batches_against_oracle():
    """Seeded sweep over frame sizes around the tile sizes of both kernels (16x32 and 18x30), batches of
    1..5 frames per launch, all four QPs, both kernel variants: every output byte equals the oracle's."""
    rng = np.random.default_rng(20250111)
    lib = capi.load()
    shapes = [(int(rng.integers(1, 120)), int(rng.integers(1, 200))) for _ in range(14)]
    shapes += [(32, 64), (36, 60), (33, 65), (35, 59), (16, 31), (54, 90), (160, 64), (17, 640)]
    for k, (h, w) in enumerate(shapes):
        qp = (22, 27, 32, 37)[k % 4]
        nf = 1 + k % 5
        mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
        frames = synth.synth_frames(h, w, qp, frames=nf, seed=1000 + k)[1]
        want = [c_oracle.forward_blu(mb, np.ascontiguousarray(frames[i])) for i in range(nf)]
        d_in, d_out = C.c_void_p(), C.c_void_p()
        capi.check(lib.qcnn_dev_alloc(C.byref(d_in), nf * h * w))
        capi.check(lib.qcnn_dev_alloc(C.byref(d_out), nf * h * w))
        capi.check(lib.qcnn_dev_h2d(d_in, np.ascontiguousarray(frames).ctypes.data, nf * h * w))
        with qvrcnn(0, 1, 1, h, w) as net:
            for exact in ((False, True) if k % 3 == 0 else (False, 2) if k % 3 == 1 else (False,)):
                net.set_exact(exact)
                net.load_static_para(mb)
                capi.check(lib.qcnn_dev_memset(d_out, 0x5A, nf * h * w))
                net.forward_blu_batch(d_in.value, d_out.value, nf, h * w)
                net.synchronize()
                out = np.empty((nf, h, w), np.uint8)
                capi.check(lib.qcnn_dev_d2h(out.ctypes.data, d_out, nf * h * w))
                for i in range(nf):
                    bad = np.argwhere(out[i] != want[i])
                    assert bad.size == 0, f"{h}x{w} qp{qp} exact={exact} frame {i}: differs at {bad[:4].tolist()}"
        capi.check(lib.qcnn_dev_free(d_in))
        capi.check(lib.qcnn_dev_free(d_out))


def test_tile_independence_property():
    """Size-independent property: the output at a pixel depends only on its 13x13 neighbourhood, so a
    crop taken 6 pixels inside a larger frame equals the same region of the full result."""
    qp, h, w = 32, 200, 333
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    y = _frame(h, w, qp)
    with qvrcnn(0, 1, 1, h, w) as net:
        net.load_static_para(mb)
        net.load_data(y)
        net.forward_blu()
        full = net.download()
    y0, x0, ch, cw = 37, 51, 90, 140
    crop = np.ascontiguousarray(y[y0:y0 + ch, x0:x0 + cw])
    with qvrcnn(0, 1, 1, ch, cw) as net:
        net.load_static_para(mb)
        net.load_data(crop)
        net.forward_blu()
        part = net.download()
    assert np.array_equal(part[6:-6, 6:-6], full[y0 + 6:y0 + ch - 6, x0 + 6:x0 + cw - 6])


def test_layer_ops_on_raw_pointers():
    """qcnn_op_* : the reference's layer API on caller-owned device buffers."""
    lib = capi.load()
    h, w, cin, cout, ks = 23, 37, 48, 16, 3
    rng = np.random.default_rng(5)
    x = rng.integers(0, 128, (cin // 4, h, w, 4), dtype=np.int8)
    wt = rng.integers(-128, 128, (cout, cin // 4, ks, ks, 4), dtype=np.int8)
    b = rng.integers(-3000, 3000, cout).astype(np.int32)
    u_ref = np.empty((cout, h, w), np.float32)
    c_oracle.lib().orc_conv_static(x.ctypes.data, cin, h, w, wt.ctypes.data, b.ctypes.data, cout, ks,
                                   u_ref.ctypes.data)
    v_ref = np.empty((cout // 4, h, w, 4), np.int8)
    c_oracle.lib().orc_quant_blu(u_ref.ctypes.data, v_ref.ctypes.data, h * w, cout, 7426, 281, 14)

    def dev(a):
        p = C.c_void_p()
        capi.check(lib.qcnn_dev_alloc(C.byref(p), a.nbytes))
        capi.check(lib.qcnn_dev_h2d(p, a.ctypes.data, a.nbytes))
        return p

    bf = b.astype(np.float32)
    dx, dw, db = dev(x), dev(wt), dev(bf)
    du, dv = dev(np.zeros_like(u_ref)), dev(np.zeros_like(v_ref))
    capi.check(lib.qcnn_op_conv_static(dx, dw, db, du, h, w, cin, cout, ks, None))
    capi.check(lib.qcnn_op_quant_blu(du, dv, h * w, cout, 7426, 281, 14, None))
    capi.check(lib.qcnn_dev_sync())
    u = np.empty_like(u_ref)
    v = np.empty_like(v_ref)
    capi.check(lib.qcnn_dev_d2h(u.ctypes.data, du, u.nbytes))
    capi.check(lib.qcnn_dev_d2h(v.ctypes.data, dv, v.nbytes))
    assert np.array_equal(u, u_ref) and np.array_equal(v, v_ref)
    for p in (dx, dw, db, du, dv):
        capi.check(lib.qcnn_dev_free(p))
    assert lib.qcnn_op_conv_static(dx, dw, db, du, h, w, cin, cout, 7, None) == -6     # unsupported ksize


def test_error_reporting():
    lib = capi.load()
    with qvrcnn(0, 1, 1, 8, 8) as net:
        with pytest.raises(capi.QcnnError) as ei:
            net.forward_blu()                             # no model loaded
        assert ei.value.code == -5
        with pytest.raises(capi.QcnnError) as ei:
            net.load_static_para("/nonexistent/model.data")
        assert ei.value.code == -2 and "cannot open model file" in str(ei.value)
        bad = bytearray(synth.synth_model(37).to_nchw_vect_c_bytes())
        bad[-4:] = (0).to_bytes(4, "little")              # C4 shift = 0
        with pytest.raises(capi.QcnnError) as ei:
            net.load_static_para(bytes(bad))
        assert ei.value.code == -3
    h = C.c_void_p()
    assert lib.qcnn_create(0, 2, 1, 8, 8, C.byref(h)) == -6      # batch != 1
    assert lib.qcnn_create(99, 1, 1, 8, 8, C.byref(h)) == -1     # no such device


@pytest.mark.parametrize("form", ["tiled", "one_tile", "two_tiles"])
@pytest.mark.parametrize("exact", [0, 1])
def test_every_form_of_kernel_b_is_selectable(form, exact, monkeypatch):
    """The three forms of kernel B (read from the environment when an engine is created: QCNN_B_TILED=1 the tiled
    form, QCNN_B_WIDE=0 the streaming form with one column tile per wave, default the streaming form with two) on the
    bordered conc1 image, with integer epilogues and with the literal float pipeline: the bytes of the oracle, on
    batches of frames with ragged edges (widths below, at and above one strip of 14 / 30 columns, single pixels)."""
    qp = 32
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    monkeypatch.setenv("QCNN_B_TILED", "1" if form == "tiled" else "0")
    monkeypatch.setenv("QCNN_B_WIDE", "0" if form == "one_tile" else "1")
    lib = capi.load()
    for h, w, nf in ((77, 131, 3), (5, 7, 2), (1, 1, 1), (33, 61, 2), (40, 30, 1), (64, 29, 1), (23, 14, 2), (20, 95, 1)):
        frames = synth.synth_frames(h, w, qp, frames=nf, seed=5)[1]
        want = [c_oracle.forward_blu(mb, np.ascontiguousarray(frames[i])) for i in range(nf)]
        d_in, d_out = C.c_void_p(), C.c_void_p()
        capi.check(lib.qcnn_dev_alloc(C.byref(d_in), nf * h * w))
        capi.check(lib.qcnn_dev_alloc(C.byref(d_out), nf * h * w))
        capi.check(lib.qcnn_dev_h2d(d_in, np.ascontiguousarray(frames).ctypes.data, nf * h * w))
        with qvrcnn(0, 1, 1, h, w) as net:
            net.set_exact(exact)
            net.load_static_para(mb)
            assert net.arith_paths == ((1, 1) if exact else (0, 0))
            capi.check(lib.qcnn_dev_memset(d_out, 0x5A, nf * h * w))
            net.forward_blu_batch(d_in.value, d_out.value, nf, h * w)
            net.synchronize()
            out = np.empty((nf, h, w), np.uint8)
            capi.check(lib.qcnn_dev_d2h(out.ctypes.data, d_out, nf * h * w))
            for i in range(nf):
                assert np.array_equal(out[i], want[i]), f"{form} exact={exact} {w}x{h} frame {i}"
        capi.check(lib.qcnn_dev_free(d_in))
        capi.check(lib.qcnn_dev_free(d_out))


@pytest.mark.parametrize("wide", ["0", "1"])
@pytest.mark.parametrize("seg_rows", [1, 7, 24, 1000])
def test_streaming_kernel_any_band_height(seg_rows, wide, monkeypatch):
    """The streaming kernel B (both forms: one and two column tiles per wave) cuts the frame into bands of rows
    (QCNN_SEG_ROWS overrides its own choice): bands of 1, 7, 24 rows and one band for the whole frame give the oracle's
    bytes, on a frame narrower than one strip (14 / 30 columns), frames of exactly three strips and ragged ones."""
    qp = 27
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    monkeypatch.setenv("QCNN_SEG_ROWS", str(seg_rows))
    monkeypatch.setenv("QCNN_B_WIDE", wide)
    for (h, w) in ((9, 5), (50, 42), (61, 99), (37, 90), (44, 121)):
        y = _frame(h, w, qp, seed=h + w)
        rec_o, taps_o = c_oracle.forward_blu(mb, y, True)
        with qvrcnn(0, 1, 1, h, w) as net:
            net.load_static_para(mb)
            _check_fused(net, y, rec_o, taps_o["conc1"])
