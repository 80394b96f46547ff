"""The QCNN_GPU driver program and the reference-shaped C++ API, run as the reference's harness runs
them (positional argv, log.txt, recon_psnr.data), against the oracle.  Bit-exact / last decimal."""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from oracle import c_oracle
from qcnn_gpu_amd import formats, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "qcnn_gpu_amd", "bin", "QCNN_GPU")
FACADE = os.path.join(ROOT, "qcnn_gpu_amd", "bin", "facade_check")


def _write_case(tmp, h, w, qps, frames=1):
    ori = None
    for qp in qps:
        o, a = synth.synth_frames(h, w, qp, frames=frames, seed=77)
        if ori is None:
            ori = synth.synth_frames(h, w, qps[0], frames=frames, seed=77)[0]
            formats.write_yuv420_luma(str(tmp / "seq.yuv"), ori, chroma_value=128)
        # anchors of every QP are degraded versions of the same original
        a = np.clip(ori.astype(np.int16) + (a.astype(np.int16) - o.astype(np.int16)), 0, 255).astype(np.uint8)
        formats.write_yuv420_luma(str(tmp / f"anchor_Q{qp}.yuv"), a, chroma_value=77)
        with open(tmp / f"model_{qp}.data", "wb") as f:
            f.write(synth.synth_model(qp).to_nchw_vect_c_bytes())
    return ori


def _run(tmp, h, w, env):
    e = dict(os.environ, QCNN_MODEL=str(tmp / "model_%d.data"), **env)
    return subprocess.run([EXE, str(tmp / "seq.yuv"), str(tmp / "anchor_"), str(h), str(w)], cwd=tmp, env=e,
                          capture_output=True, text=True, timeout=300)


def test_driver_single_qp_outputs(tmp_path):
    h, w, qp = 240, 416, 37
    ori = _write_case(tmp_path, h, w, [qp])
    r = _run(tmp_path, h, w, {"QCNN_QPS": str(qp), "QCNN_SAVE_RECON": "recon.yuv", "QCNN_PSNR_PF": "1"})
    assert r.returncode == 0, r.stdout + r.stderr
    anc = formats.read_yuv420_luma(str(tmp_path / f"anchor_Q{qp}.yuv"), 1, h, w)
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    rec_o = c_oracle.forward_blu(mb, anc[0])
    p1, p2 = c_oracle.psnr(anc[0], ori[0]), c_oracle.psnr(rec_o, ori[0])
    m = re.search(r"\nbefore net:PSNR=([\d.]+)\nafter quantized net:PSNR=([\d.]+)\ntime:(\d+)us\n", r.stdout)
    assert m, r.stdout
    assert m.group(1) == "%.3f" % p1 and m.group(2) == "%.3f" % p2          # last printed decimal
    assert ("PSNR of Frame 0:%f" % p2) in r.stdout
    # recon_psnr.data: one raw double, identical bits
    raw = (tmp_path / "recon_psnr.data").read_bytes()
    assert raw == struct.pack("<d", p2)
    # log.txt record
    log = (tmp_path / "log.txt").read_text()
    assert log.startswith("\nQVRCNN test date:")
    assert ("data:%s\nframes:1\nheight:%d\nwidth:%d\nbefore net:PSNR=%f\nafter quantized net:PSNR=%f\ntime:%sus\n"
            % (str(tmp_path / f"anchor_Q{qp}.yuv"), h, w, p1, p2, m.group(3))) in log
    # reconstruction file: luma bit-exact, chroma zero-filled
    rec = formats.read_yuv420_luma(str(tmp_path / "recon.yuv"), 1, h, w)
    assert np.array_equal(rec[0], rec_o)
    assert (tmp_path / "recon.yuv").stat().st_size == h * w * 3 // 2


def test_driver_pipelined_clip_logs_per_frame_psnr(tmp_path):
    """configs[4] in miniature: a 21-frame clip at QP27 through QCNN_PIPELINE=1 (H2D / kernels / D2H
    overlapped, PSNR sums on the device); per-frame and whole-clip PSNR equal the oracle's doubles and the
    blocking loop's outputs byte for byte."""
    h, w, qp, nf = 72, 120, 27, 21
    ori = _write_case(tmp_path, h, w, [qp], frames=nf)
    env = {"QCNN_QPS": str(qp), "QCNN_FRAMES": str(nf), "QCNN_PSNR_PF": "1", "QCNN_SAVE_RECON": "recon.yuv"}
    r = _run(tmp_path, h, w, dict(env, QCNN_PIPELINE="1", QCNN_BATCH_GROUP="4"))
    assert r.returncode == 0, r.stdout + r.stderr
    anc = formats.read_yuv420_luma(str(tmp_path / f"anchor_Q{qp}.yuv"), nf, h, w)
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    rec_o = np.stack([c_oracle.forward_blu(mb, np.ascontiguousarray(anc[i])) for i in range(nf)])
    for i in range(nf):
        assert ("PSNR of Frame %d:%f\n" % (i, c_oracle.psnr(rec_o[i], np.ascontiguousarray(ori[i])))) in r.stdout
    m = re.search(r"\nbefore net:PSNR=([\d.]+)\nafter quantized net:PSNR=([\d.]+)\ntime:(\d+)us\n", r.stdout)
    assert m, r.stdout
    assert m.group(1) == "%.3f" % c_oracle.psnr(anc, ori) and m.group(2) == "%.3f" % c_oracle.psnr(rec_o, ori)
    rec = formats.read_yuv420_luma(str(tmp_path / "recon.yuv"), nf, h, w)
    assert np.array_equal(rec, rec_o)
    pipelined = (tmp_path / "recon_psnr.data").read_bytes()
    # the blocking loop of the reference gives the same record
    (tmp_path / "recon_psnr.data").unlink()
    r2 = _run(tmp_path, h, w, env)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    assert (tmp_path / "recon_psnr.data").read_bytes() == pipelined
    strip = lambda t: re.sub(r"time:\d+us", "", t)
    assert strip(r.stdout) == strip(r2.stdout)


def test_driver_four_qps_and_frames_in_one_process(tmp_path):
    """configs[2] at the process level: QP 22,27,32,37 back to back, 3 frames each, appended outputs."""
    h, w, qps, frames = 96, 160, [22, 27, 32, 37], 3
    ori = _write_case(tmp_path, h, w, qps, frames)
    r = _run(tmp_path, h, w, {"QCNN_QPS": "22,27,32,37", "QCNN_FRAMES": str(frames)})
    assert r.returncode == 0, r.stdout + r.stderr
    doubles = formats.read_psnr_doubles(str(tmp_path / "recon_psnr.data"))
    assert doubles.shape == (4,)
    for i, qp in enumerate(qps):
        anc = formats.read_yuv420_luma(str(tmp_path / f"anchor_Q{qp}.yuv"), frames, h, w)
        mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
        rec = np.stack([c_oracle.forward_blu(mb, anc[f]) for f in range(frames)])
        assert doubles[i] == c_oracle.psnr(rec, ori)             # whole-clip PSNR, identical double
    assert (tmp_path / "log.txt").read_text().count("QVRCNN test date:") == 4


def test_driver_errors_like_the_reference(tmp_path):
    _write_case(tmp_path, 32, 32, [37])
    r = _run(tmp_path, 32, 32, {"QCNN_QPS": "27"})                # no model / anchor for QP27
    assert r.returncode == 1 and "cannot open model file." in r.stdout
    e = dict(os.environ, QCNN_MODEL=str(tmp_path / "model_%d.data"), QCNN_QPS="37")
    r = subprocess.run([EXE, str(tmp_path / "missing.yuv"), str(tmp_path / "anchor_"), "32", "32"], cwd=tmp_path,
                       env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "open ori file failed" in r.stdout


def test_driver_model_conversion(tmp_path):
    m = synth.synth_model(32)
    (tmp_path / "hwcn_32.data").write_bytes(m.to_hwcn_bytes())
    e = dict(os.environ, QCNN_CONVERT="1", QCNN_QPS="32", QCNN_MODEL_HWCN=str(tmp_path / "hwcn_%d.data"),
             QCNN_MODEL=str(tmp_path / "vect_%d.data"))
    r = subprocess.run([EXE], cwd=tmp_path, env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "vect_32.data").read_bytes() == m.to_nchw_vect_c_bytes()


def test_cpp_layer_api_matches_oracle(tmp_path):
    h, w, qp = 50, 70, 27
    ori = _write_case(tmp_path, h, w, [qp])
    out = tmp_path / "out"
    r = subprocess.run([FACADE, str(tmp_path / f"model_{qp}.data"), str(tmp_path / "seq.yuv"),
                        str(tmp_path / f"anchor_Q{qp}.yuv"), str(h), str(w), str(out)],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, FACADE_VIEWMEM="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    anc = formats.read_yuv420_luma(str(tmp_path / f"anchor_Q{qp}.yuv"), 1, h, w)
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    rec_o, taps = c_oracle.forward_blu(mb, anc[0], True)
    assert np.array_equal(np.fromfile(str(out) + ".fused", np.uint8).reshape(h, w), rec_o)
    assert np.array_equal(np.fromfile(str(out) + ".layers", np.uint8).reshape(h, w), rec_o)
    assert np.array_equal(np.fromfile(str(out) + ".c1v", np.int8).reshape(16, h, w, 4), taps["v1"])
    t = synth.QUANT_TRIPLES[qp]
    assert f"C1 blu={t[0][0]} mul={t[0][1]} shift={t[0][2]}  C4 mul={t[5][1]} shift={t[5][2]}" in r.stdout
    assert "mul:%d,shift:%d" % (t[0][1], t[0][2]) in r.stdout      # viewmem dump, inference/cnn.cu:235
    # members the reference declares without a body (inference/cnn.cuh:33-34,69, qvrcnn.cuh:33) and InputLayer::viewmem
    # (inference/cnn.cu:536-549): freeMem / setMem release and restore u and v of an owning layer and refuse on an
    # adopted one, concat_static does its host bookkeeping and reports unsupported (-6), viewmem is silent
    step_y = ((1 << t[1][2]) + t[1][1] // 2) // t[1][1]
    assert ("declared-only members: freeMem 0 (1) setMem 0 (1) adopted freeMem -6 concat_static -6 step_y %d viewmem 0 "
            "backward -6" % step_y) in r.stdout


def test_cpp_dynamic_layer_api_matches_oracle(tmp_path):
    """The dynamic-quantisation members of the C++ classes (qvrcnn::load_para / forward / save_b_adj,
    CovLayer::load_para / ConvForward / activate / quantize_out, ConcatLayer::concat, adjustBasic /
    adjustOutput, InputLayer::applyRes) against the oracle of inference/qvrcnn.cu:82-167."""
    h, w, qp = 50, 70, 27
    _write_case(tmp_path, h, w, [qp])
    dm = synth.synth_dynamic_model(qp)
    (tmp_path / "dyn.data").write_bytes(dm.to_bytes())
    out = tmp_path / "out"
    r = subprocess.run([FACADE, str(tmp_path / f"model_{qp}.data"), str(tmp_path / "seq.yuv"),
                        str(tmp_path / f"anchor_Q{qp}.yuv"), str(h), str(w), str(out), str(tmp_path / "dyn.data")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    anc = formats.read_yuv420_luma(str(tmp_path / f"anchor_Q{qp}.yuv"), 1, h, w)
    rec_o, t = c_oracle.forward_dynamic(dm.to_bytes(), anc[0], True)
    assert np.array_equal(np.fromfile(str(out) + ".dyn", np.uint8).reshape(h, w), rec_o)
    assert np.array_equal(np.fromfile(str(out) + ".dynlayers", np.uint8).reshape(h, w), rec_o)
    assert np.array_equal(np.fromfile(str(out) + ".c1dyn", np.int8).reshape(16, h, w, 4), t["v1"])
    assert np.array_equal(np.fromfile(str(out) + ".b_adj", np.float32), t["b_adj"])
    steps = "w %d %d %d %d y %d %d %d %d" % (*t["stepw"], *t["stepy"])
    assert "dyn steps " + steps in r.stdout and "layers steps " + steps in r.stdout
    sy, mu = t["step_y"], t["max_u"]
    assert "dyn step_y %d %d %d %d %d max_u %d %d" % (sy[0], sy[1], sy[2], sy[3], sy[4], mu[0], mu[4]) in r.stdout
    assert "standalone step_w %d step_y %d max_u %d" % (dm.layers[0].step_w, sy[0], mu[0]) in r.stdout
    # qvrcnn::forward as its shipped text reads (quantize_out_static, concat_blu, adjustOutput_static(.., 141, 16) inside
    # the dynamic pass, inference/qvrcnn.cu:100,122,141,157) against the oracle's literal restatement
    rec_s, stepw, stepy, step_y = c_oracle.forward_shipped(dm.to_bytes(), synth.QUANT_TRIPLES[qp], 37, anc[0])
    assert np.array_equal(np.fromfile(str(out) + ".shipped", np.uint8).reshape(h, w), rec_s)
    assert ("shipped steps w %d %d %d %d y %d %d %d %d step_y %d %d %d %d" % (*stepw, *stepy, *step_y[1:5])) in r.stdout
    assert not np.array_equal(rec_s, rec_o)          # it is a different pass, not an alias of forward()


def test_driver_dynamic_mode(tmp_path):
    """QCNN_DYNAMIC=1: the driver's other mode (inference/kernel.cu:87,93 commented alternative,
    load_para + forward) over a 3-frame clip; PSNR equals the oracle's to the last printed decimal."""
    h, w, qp, frames = 64, 96, 32, 3
    ori, anc = synth.synth_frames(h, w, qp, frames=frames)
    formats.write_yuv420_luma(str(tmp_path / "seq.yuv"), ori)
    formats.write_yuv420_luma(str(tmp_path / f"anchor_Q{qp}.yuv"), anc)
    dm = synth.synth_dynamic_model(qp).to_bytes()
    (tmp_path / f"model_{qp}.data").write_bytes(dm)
    r = _run(tmp_path, h, w, {"QCNN_QPS": str(qp), "QCNN_FRAMES": str(frames), "QCNN_DYNAMIC": "1",
                              "QCNN_SAVE_RECON": "recon.yuv"})
    assert r.returncode == 0, r.stdout + r.stderr
    rec_o = np.stack([c_oracle.forward_dynamic(dm, a) for a in anc])
    m = re.search(r"\nbefore net:PSNR=([\d.]+)\nafter quantized net:PSNR=([\d.]+)\ntime:(\d+)us\n", r.stdout)
    assert m, r.stdout
    assert m.group(1) == "%.3f" % c_oracle.psnr(anc, ori) and m.group(2) == "%.3f" % c_oracle.psnr(rec_o, ori)
    assert np.array_equal(formats.read_yuv420_luma(str(tmp_path / "recon.yuv"), frames, h, w), rec_o)


def test_driver_float_mode(tmp_path):
    """QCNN_FLOAT=1: the driver as the reference built with FLOAT_CONFIG runs it (inference/kernel.cu:13-14,
    NCHW float model, load_para + forward of inference/qvrcnn.cu:244-285) over a 2-frame clip.  Tolerance: the
    reconstructed pixels may differ from the double-accumulating oracle by one grey level on at most 0.1 % of
    them (float32 summation order), so the printed PSNR may differ in its last decimal by at most 1."""
    h, w, qp, frames = 64, 96, 27, 2
    ori, anc = synth.synth_frames(h, w, qp, frames=frames)
    formats.write_yuv420_luma(str(tmp_path / "seq.yuv"), ori)
    formats.write_yuv420_luma(str(tmp_path / f"anchor_Q{qp}.yuv"), anc)
    fm = synth.synth_float_model(qp)
    (tmp_path / f"model_{qp}.data").write_bytes(fm)
    r = _run(tmp_path, h, w, {"QCNN_QPS": str(qp), "QCNN_FRAMES": str(frames), "QCNN_FLOAT": "1",
                              "QCNN_SAVE_RECON": "recon.yuv"})
    assert r.returncode == 0, r.stdout + r.stderr
    rec_o = np.stack([c_oracle.forward_float(fm, a) for a in anc])
    rec = formats.read_yuv420_luma(str(tmp_path / "recon.yuv"), frames, h, w)
    d = np.abs(rec.astype(int) - rec_o.astype(int))
    assert d.max() <= 1 and np.count_nonzero(d) <= max(1, rec.size // 1000)
    m = re.search(r"\nbefore net:PSNR=([\d.]+)\nafter quantized net:PSNR=([\d.]+)\ntime:(\d+)us\n", r.stdout)
    assert m, r.stdout
    assert m.group(1) == "%.3f" % c_oracle.psnr(anc, ori)
    assert m.group(2) == "%.3f" % c_oracle.psnr(rec, ori)
    assert abs(float(m.group(2)) - c_oracle.psnr(rec_o, ori)) <= 0.0015


def test_driver_chroma_extension(tmp_path):
    """QCNN_CHROMA=1: after the luma pass the driver runs the same network over the half-size U and V planes of
    the 4:2:0 files and patches them into the reconstruction file (the reference skips chroma and writes zeros,
    inference/yuv_data.cpp:36-37,119-125).  Bit-exact against the oracle run on every plane; the luma records on
    stdout, in log.txt and recon_psnr.data are unchanged by the extension."""
    h, w, qp, frames = 48, 64, 37, 2
    ori, anc = synth.synth_frames(h, w, qp, frames=frames)
    # chroma planes: synthetic half-size frames with their own seeds (planes 2i = U, 2i+1 = V of frame i)
    cori, canc = synth.synth_frames(h // 2, w // 2, qp, frames=2 * frames, seed=0xC0FFEE)
    for name, lum, chr_ in (("seq.yuv", ori, cori), (f"anchor_Q{qp}.yuv", anc, canc)):
        with open(tmp_path / name, "wb") as f:
            for i in range(frames):
                f.write(lum[i].tobytes() + chr_[2 * i].tobytes() + chr_[2 * i + 1].tobytes())
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    (tmp_path / f"model_{qp}.data").write_bytes(mb)
    r = _run(tmp_path, h, w, {"QCNN_QPS": str(qp), "QCNN_FRAMES": str(frames), "QCNN_CHROMA": "1",
                              "QCNN_SAVE_RECON": "recon.yuv"})
    assert r.returncode == 0, r.stdout + r.stderr
    rec_y = np.stack([c_oracle.forward_blu(mb, a) for a in anc])
    rec_c = np.stack([c_oracle.forward_blu(mb, a) for a in canc])
    m = re.search(r"\nbefore net:PSNR=([\d.]+)\nafter quantized net:PSNR=([\d.]+)\ntime:(\d+)us\n", r.stdout)
    assert m and m.group(1) == "%.3f" % c_oracle.psnr(anc, ori) and m.group(2) == "%.3f" % c_oracle.psnr(rec_y, ori)
    m = re.search(r"chroma before net:PSNR=([\d.]+)\nchroma after quantized net:PSNR=([\d.]+)\nchroma time:(\d+)us\n", r.stdout)
    assert m, r.stdout
    assert m.group(1) == "%.3f" % c_oracle.psnr(canc, cori) and m.group(2) == "%.3f" % c_oracle.psnr(rec_c, cori)
    raw = np.fromfile(tmp_path / "recon.yuv", np.uint8).reshape(frames, h * w * 3 // 2)
    q = h * w // 4
    for i in range(frames):
        assert np.array_equal(raw[i, :h * w].reshape(h, w), rec_y[i])
        assert np.array_equal(raw[i, h * w:h * w + q].reshape(h // 2, w // 2), rec_c[2 * i])
        assert np.array_equal(raw[i, h * w + q:].reshape(h // 2, w // 2), rec_c[2 * i + 1])
    assert (tmp_path / "recon_psnr.data").read_bytes() == struct.pack("<d", c_oracle.psnr(rec_y, ori))


@pytest.mark.parametrize("pipeline", [False, True])
@pytest.mark.parametrize("workers", [2, 3])
def test_driver_multi_gpu_workers_on_one_device(tmp_path, pipeline, workers):
    """The thread-per-GPU sharding of the driver (frame i -> worker i mod G, one engine and one host thread per
    worker, no communication; the multi-GPU half of configs[3] / configs[4]) with QCNN_GPU_MAP folding every
    worker onto device 0, so that it runs on a one-GPU box: reconstruction, per-frame PSNR lines,
    recon_psnr.data and the stdout record equal the one-worker run byte for byte, and the oracle."""
    h, w, qp, nf = 72, 120, 27, 11          # 11 frames: ragged over 2 and over 3 workers
    ori = _write_case(tmp_path, h, w, [qp], frames=nf)
    env = {"QCNN_QPS": str(qp), "QCNN_FRAMES": str(nf), "QCNN_PSNR_PF": "1", "QCNN_SAVE_RECON": "recon.yuv",
           "QCNN_BATCH_GROUP": "2"}
    if pipeline:
        env["QCNN_PIPELINE"] = "1"
    r1 = _run(tmp_path, h, w, env)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    single_rec = (tmp_path / "recon.yuv").read_bytes()
    single_psnr = (tmp_path / "recon_psnr.data").read_bytes()
    (tmp_path / "recon_psnr.data").unlink()
    (tmp_path / "recon.yuv").unlink()
    rN = _run(tmp_path, h, w, dict(env, QCNN_GPUS=str(workers), QCNN_GPU_MAP="0"))
    assert rN.returncode == 0, rN.stdout + rN.stderr
    assert (tmp_path / "recon.yuv").read_bytes() == single_rec
    assert (tmp_path / "recon_psnr.data").read_bytes() == single_psnr
    strip = lambda t: re.sub(r"time:\d+us", "", t)
    assert strip(rN.stdout) == strip(r1.stdout)
    assert rN.stdout.count("PSNR of Frame") == nf
    anc = formats.read_yuv420_luma(str(tmp_path / f"anchor_Q{qp}.yuv"), nf, h, w)
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    rec_o = np.stack([c_oracle.forward_blu(mb, np.ascontiguousarray(anc[i])) for i in range(nf)])
    assert np.array_equal(formats.read_yuv420_luma(str(tmp_path / "recon.yuv"), nf, h, w), rec_o)
    assert single_psnr == struct.pack("<d", c_oracle.psnr(rec_o, ori))


def test_driver_gpu_map_rejects_missing_device(tmp_path):
    _write_case(tmp_path, 32, 32, [37])
    r = _run(tmp_path, 32, 32, {"QCNN_QPS": "37", "QCNN_GPUS": "2", "QCNN_GPU_MAP": "0,99"})
    assert r.returncode == 1 and "QCNN_GPU_MAP names device 99" in r.stdout


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (here both on device 0) and rank 0
    prints one JSON line with n_gpus = 2; without --allow-shared-gpus the same command refuses to run."""
    import json
    import sys
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--height", "90",
            "--width", "160", "--batch", "8", "--prewarm", "0", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(base + ["--allow-shared-gpus"], capture_output=True, text=True, timeout=600, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["parity_vs_oracle"] is True and out["scaling"] == "weak"
    assert out["value"] > 0 and "roofline" in out
    # one entry per rank: its own rate, and the device it ran on (both ranks on the one card here)
    assert len(out["per_rank_mpixels_s"]) == 2 and all(v > 0 for v in out["per_rank_mpixels_s"])
    assert len(out["per_rank_device"]) == 2 and len(out["per_rank_kernel_us"]) == 2
    assert out["parity_frames_checked"] == 16            # every frame of both ranks
    n = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True)
    if n.stdout.strip() == "1":
        r = subprocess.run(base, capture_output=True, text=True, timeout=600, env=env, cwd=tmp_path)
        assert r.returncode != 0 and "only 1 GPU(s) present" in r.stderr
