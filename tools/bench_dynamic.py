#!/usr/bin/env python3
"""Times the dynamic-quantisation pass (qcnn_forward, qvrcnn::forward of inference/qvrcnn.cu:82-167) and,
beside it, the layer-by-layer static pass (qcnn_forward_blu_layered) and the float32 configuration
(qcnn_forward_float, inference/qvrcnn.cu:244-285) on one resident 1080p Y frame.
Prints one JSON line.  Not the headline benchmark (bench.py is); this path is bound by the float32
intermediates it has to write for the per-frame maxima: 4 B per output channel and pixel.

    python tools/bench_dynamic.py [--steps K] [--warmup W] [--height H] [--width W] [--qp QP]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qcnn_gpu_amd import synth  # noqa: E402
from qcnn_gpu_amd.qvrcnn import qvrcnn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--qp", type=int, default=37)
ap.add_argument("--only", default="", help="comma-separated subset of: dynamic,dynamic_layered,static_layered,float32")
a = ap.parse_args()
h, w, qp = a.height, a.width, a.qp
y = synth.synth_frames(h, w, qp)[1][0]
out = {"workload": f"{w}x{h} Y frame, QP{qp}, resident in HBM", "steps": a.steps}
with qvrcnn(0, 1, 1, h, w) as net:
    for name, load, run in (("dynamic", lambda: net.load_para(synth.synth_dynamic_model(qp).to_bytes()), net.forward),
                            ("dynamic_layered", lambda: None, net.forward_layered),
                            ("static_layered", lambda: net.load_static_para(synth.synth_model(qp).to_nchw_vect_c_bytes()),
                             net.forward_blu_layered),
                            ("float32", lambda: net.load_float_para(synth.synth_float_model(qp)), net.forward_float)):
        if a.only and name not in a.only.split(","):
            if name == "dynamic":
                load()  # the layered variant runs on the model this one loads
            continue
        load()
        net.load_data(y)
        for _ in range(a.warmup):
            run()
        net.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            run()
        net.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        out[name] = {"ms_per_frame": round(dt * 1e3, 3), "mpixels_s": round(h * w / dt / 1e6, 1)}
    if os.environ.get("CHECK", "1") != "0":
        from oracle import c_oracle
        dm = synth.synth_dynamic_model(qp).to_bytes()
        net.load_para(dm)
        net.load_data(y)
        net.forward()
        out["parity_vs_oracle"] = bool(np.array_equal(net.download(), c_oracle.forward_dynamic(dm, y)))
print(json.dumps(out))
