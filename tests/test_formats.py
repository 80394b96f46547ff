"""File formats of the reference: model images, quant_params pickles (parsed without unpickling),
PSNR double files, log/stdout records, YUV 4:2:0 luma."""
import math
import os
import struct

import numpy as np
import pytest

from qcnn_gpu_amd import formats, synth

# SURVEY Appendix C column heads: first three 'after net' PSNRs of each psnr_static_%d.data
PSNR_HEAD = {22: (43.559227, 43.250319, 43.174014), 27: (40.460927, 39.007978, 41.919759),
             32: (37.449883, 34.975122, 40.166436), 37: (34.466063, 32.845786, 37.733835)}


@pytest.mark.parametrize("qp", [22, 27, 32, 37])
def test_quant_params_pickle_is_parsed_without_unpickling(qp, golden_dir):
    rows = formats.read_quant_params(os.path.join(golden_dir, f"quant_params{qp}.data"))
    assert len(rows) == 6 and all(len(r) == 6 for r in rows)
    assert tuple((int(r[3]), int(r[4]), int(r[5])) for r in rows) == synth.QUANT_TRIPLES[qp]
    assert rows[0][1] == 255                                  # C1 ratio
    # output-scale chain: ratio' = ratio / stepw * mul / 2^shift (training/quantization.py:58-62)
    r1 = rows[0][1] / rows[0][0] * rows[0][4] / 2 ** rows[0][5]
    assert math.isclose(r1, rows[1][1], rel_tol=1e-12) and rows[1][1] == rows[2][1]
    r2 = rows[1][1] / rows[1][0] * rows[1][4] / 2 ** rows[1][5]
    assert math.isclose(r2, rows[3][1], rel_tol=1e-12) and rows[3][1] == rows[4][1]


def test_quant_params_parser_rejects_foreign_globals():
    evil = b"\x80\x03cos\nsystem\nq\x00X\x02\x00\x00\x00idq\x01\x85q\x02Rq\x03."
    with pytest.raises(ValueError):
        formats.parse_quant_params_pickle(evil)


def test_quant_params_cpp_roundtrip(tmp_path, golden_dir):
    rows = formats.read_quant_params(os.path.join(golden_dir, "quant_params37.data"))
    p = tmp_path / "quant_params_cpp_37.data"
    formats.write_quant_params_cpp(str(p), rows)
    assert p.stat().st_size == 288
    back = formats.read_quant_params_cpp(str(p))
    assert back == [[float(v) for v in r] for r in rows]


# The "after net" PSNR of the 18 sequences of inference/run_all.bat:2-23, in that order, as the reference's own log
# prints them with %f (inference/log.txt, the 18 consecutive records from lines 19164 (QP22), 18840 (QP27),
# 19002 (QP32) and 18480 (QP37)); psnr_static_%d.data holds the same numbers as raw doubles.
PSNR_PUBLISHED = {
    22: (43.559227, 43.250319, 43.174014, 41.698177, 40.541803, 41.100642, 42.636192, 42.099443, 41.876322, 41.218068,
         42.191939, 42.826159, 41.777968, 41.306464, 42.734570, 44.102494, 44.171832, 44.654846),
    27: (40.460927, 39.007978, 41.919759, 38.916977, 37.929073, 39.458542, 37.872313, 38.800404, 38.883504, 36.778530,
         38.594294, 39.425509, 37.441326, 37.414731, 38.612918, 41.680043, 42.163152, 42.473673),
    32: (37.449883, 34.975122, 40.166436, 36.068996, 35.677459, 38.013338, 34.673137, 35.897398, 35.565503, 32.866749,
         34.947060, 36.083019, 33.797820, 33.877349, 34.645364, 38.916625, 39.958420, 40.009500),
    37: (34.466063, 32.845786, 37.733835, 33.263627, 33.344701, 36.277800, 31.794696, 33.279363, 32.156799, 29.095401,
         31.383427, 32.835384, 30.311630, 30.598423, 31.178195, 35.822853, 37.462374, 37.195164),
}


@pytest.mark.parametrize("qp", [22, 27, 32, 37])
def test_psnr_static_files(qp, golden_dir):
    d = formats.read_psnr_doubles(os.path.join(golden_dir, f"psnr_static_{qp}.data"))
    assert d.shape == (18,)
    assert np.allclose(d[:3], PSNR_HEAD[qp], atol=5e-7)
    # all 18 doubles of the file against the published six-decimal figures
    assert ["%.6f" % v for v in d] == ["%.6f" % v for v in PSNR_PUBLISHED[qp]]


def test_psnr_append_and_records(tmp_path):
    p = tmp_path / "recon_psnr.data"
    formats.append_psnr_double(str(p), 36.2778)
    formats.append_psnr_double(str(p), float("inf"))
    d = formats.read_psnr_doubles(str(p))
    assert d[0] == 36.2778 and np.isinf(d[1])
    out = formats.format_stdout_record(35.825016, 36.2778, 57443)
    assert out == "\nbefore net:PSNR=35.825\nafter quantized net:PSNR=36.278\ntime:57443us\n"
    rec = formats.format_log_record(0.0, "a_Q37.yuv", 1, 1080, 1920, 35.825016, 36.2778, 57443)
    lines = rec.split("\n")
    assert lines[0] == "" and lines[1].startswith("QVRCNN test date:")
    assert lines[2:] == ["data:a_Q37.yuv", "frames:1", "height:1080", "width:1920", "before net:PSNR=35.825016",
                         "after quantized net:PSNR=36.277800", "time:57443us", ""]


def test_model_roundtrip_and_sizes():
    m = synth.synth_model(32)
    img = m.to_nchw_vect_c_bytes()
    m2 = formats.StaticModel.from_nchw_vect_c_bytes(img)
    for a, b in zip(m.layers, m2.layers):
        assert np.array_equal(a.w, b.w) and np.array_equal(a.b, b.b)
        assert (a.blu, a.mul, a.shift) == (b.blu, b.mul, b.shift)
    assert m2.to_nchw_vect_c_bytes() == img
    with pytest.raises(ValueError):
        formats.StaticModel.from_nchw_vect_c_bytes(img[:-1])
    # last 12 bytes: C4's blu, mul, shift
    assert struct.unpack("<3i", img[-12:]) == synth.QUANT_TRIPLES[32][5]


def test_yuv_luma_reader_skips_chroma(tmp_path):
    h, w, n = 6, 8, 3
    frames = (np.arange(n * h * w) % 251).astype(np.uint8).reshape(n, h, w)
    p = tmp_path / "t.yuv"
    formats.write_yuv420_luma(str(p), frames)
    assert p.stat().st_size == n * (h * w + h * w // 2)       # luma + H*W/2 chroma bytes per frame
    got = formats.read_yuv420_luma(str(p), n, h, w)
    assert np.array_equal(got, frames)
    assert np.array_equal(formats.read_yuv420_luma(str(p), 1, h, w, start=2)[0], frames[2])
    with pytest.raises(ValueError):
        formats.read_yuv420_luma(str(p), n + 1, h, w)


def test_synthetic_generator_is_deterministic_and_in_range():
    o1, a1 = synth.synth_frames(24, 40, 37, frames=2)
    o2, a2 = synth.synth_frames(24, 40, 37, frames=2)
    assert np.array_equal(o1, o2) and np.array_equal(a1, a2)
    assert o1.min() >= 16 and o1.max() <= 235
    assert not np.array_equal(o1[0], o1[1])
    assert synth.splitmix64(0, 2).tolist() == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4]   # reference values
