#!/usr/bin/env python3
"""Writes the golden fixtures of the dynamic-quantisation pass under tests/golden/:

  dynamic_model_qp{22,37}.data   dynamic model images (step_w + weights + biases per layer, 59 980 B)
  dynamic_case_qp{QP}.npz        40x56 anchor frame, expected x_rec, the step bookkeeping and a CRC of conc2,
                                 computed by the numpy oracle and cross-checked against the C oracle here

These are this build's own vectors (the reference has neither weights nor video): they pin the two
oracles and the GPU path to each other across rounds.
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle, np_dynamic  # noqa: E402
from qcnn_gpu_amd import synth  # noqa: E402

out = os.path.join(ROOT, "tests", "golden")
for qp in (22, 37):
    m = synth.synth_dynamic_model(qp)
    mb = m.to_bytes()
    with open(os.path.join(out, f"dynamic_model_qp{qp}.data"), "wb") as f:
        f.write(mb)
    ori, anc = synth.synth_frames(40, 56, qp)
    rec, t = np_dynamic.forward(m, anc[0], True)
    rec_c, t_c = c_oracle.forward_dynamic(mb, anc[0], True)
    assert np.array_equal(rec, rec_c) and np.array_equal(t["conc2"], t_c["conc2"])
    np.savez_compressed(os.path.join(out, f"dynamic_case_qp{qp}.npz"), anchor=anc[0], rec=rec,
                        step_y=np.array(t["step_y"], np.int32), max_u=np.array(t["max_u"], np.int32),
                        stepw=np.array(t["stepw"], np.int32), stepy=np.array(t["stepy"], np.int32),
                        crc_conc2=np.uint32(zlib.crc32(np.ascontiguousarray(t["conc2"]).tobytes())))
    print(qp, "step_y", t["step_y"], "stepw", t["stepw"], "stepy", t["stepy"], "residual range", t["v4"].min(),
          t["v4"].max())
