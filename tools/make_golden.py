#!/usr/bin/env python3
"""Writes the synthetic golden fixtures under tests/golden/:

  synth_model_qp{22,27,32,37}.data   NCHW_VECT_C static model images (60 028 B each)
  synth_case_qp{QP}.npz              anchor/original 40x56 Y frames, expected x_rec and a CRC of
                                     every tap, as computed by the numpy oracle and cross-checked
                                     against the C oracle here

The reference has no weights or video, so these are this build's own vectors: they pin the two
oracles and the GPU path to each other across rounds, not to the reference.
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle, np_oracle  # noqa: E402
from qcnn_gpu_amd import synth  # noqa: E402

out = os.path.join(ROOT, "tests", "golden")
for qp in (22, 27, 32, 37):
    m = synth.synth_model(qp)
    mb = m.to_nchw_vect_c_bytes()
    with open(os.path.join(out, f"synth_model_qp{qp}.data"), "wb") as f:
        f.write(mb)
    ori, anc = synth.synth_frames(40, 56, qp)
    rec, taps = np_oracle.forward_blu(m, anc[0], True)
    rec_c, taps_c = c_oracle.forward_blu(mb, anc[0], True)
    assert np.array_equal(rec, rec_c)
    crcs = {}
    for k in ("x_ppro", "v1", "conc1", "conc2"):
        assert np.array_equal(taps[k], taps_c[k])
        crcs[k] = zlib.crc32(np.ascontiguousarray(taps[k]).tobytes())
    for l in range(6):
        assert np.array_equal(taps["u"][l], taps_c["u"][l])
        crcs[f"u{l}"] = zlib.crc32(np.ascontiguousarray(taps["u"][l]).tobytes())
    np.savez_compressed(os.path.join(out, f"synth_case_qp{qp}.npz"), ori=ori[0], anchor=anc[0], rec=rec,
                        conc1=taps["conc1"], crc_names=np.array(sorted(crcs)),
                        crc_values=np.array([crcs[k] for k in sorted(crcs)], np.uint32),
                        psnr_before=np_oracle.psnr(anc[0], ori[0]), psnr_after=np_oracle.psnr(rec, ori[0]))
    print(qp, "psnr", np_oracle.psnr(anc[0], ori[0]), "->", np_oracle.psnr(rec, ori[0]))
