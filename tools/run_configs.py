#!/usr/bin/env python3
"""Runs the clip-sized benchmark configurations through the QCNN_GPU driver on one GPU and checks a sample
of the per-frame PSNR lines against the CPU oracle:

  config 3: the four QP models back to back in one process on 1080p, model-switch cost reported
  config 5: 1920x1080 x 500 frames, QP27, per-frame PSNR logged ("PSNR of Frame %d:%f")
  config 4: 3840x2160 x 64 frames, QP32

The real sequences are not available, so the clips are synthetic (50 / 8 distinct frames, repeated); files are
written under a scratch directory and deleted.  Prints one JSON object.

    python tools/run_configs.py [--scratch DIR] [--check N]
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qcnn_gpu_amd import synth  # noqa: E402

EXE = os.path.join(ROOT, "qcnn_gpu_amd", "bin", "QCNN_GPU")


def write_clip(path, frames, repeat):
    h, w = frames.shape[1:]
    uv = bytes([128]) * (h * w // 2)
    with open(path, "wb") as f:
        for _ in range(repeat):
            for fr in frames:
                f.write(fr.tobytes())
                f.write(uv)


def run_case(tmp, name, h, w, qp, distinct, repeat, check):
    from oracle import c_oracle
    d = os.path.join(tmp, name)
    os.makedirs(d)
    t0 = time.time()
    ori, anc = synth.synth_frames(h, w, qp, frames=distinct)
    write_clip(os.path.join(d, "seq.yuv"), ori, repeat)
    write_clip(os.path.join(d, f"anchor_Q{qp}.yuv"), anc, repeat)
    mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
    with open(os.path.join(d, f"model_{qp}.data"), "wb") as f:
        f.write(mb)
    print(f"[{name}] clip written in {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    n = distinct * repeat
    out = {"frames": n, "height": h, "width": w, "qp": qp, "distinct_frames": distinct}
    for mode, env in (("pipelined", {"QCNN_PIPELINE": "1"}), ("blocking", {}), ("streaming", {"QCNN_STREAM": "1"}),
                      ("streaming_with_recon_file", {"QCNN_STREAM": "1", "QCNN_SAVE_RECON": os.path.join(d, "recon.yuv")})):
        e = dict(os.environ, QCNN_MODEL=os.path.join(d, "model_%d.data"), QCNN_QPS=str(qp), QCNN_FRAMES=str(n),
                 QCNN_PSNR_PF="1", **env)
        r = subprocess.run([EXE, os.path.join(d, "seq.yuv"), os.path.join(d, "anchor_"), str(h), str(w)], cwd=d, env=e,
                           capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            raise RuntimeError(r.stdout[-2000:] + r.stderr[-2000:])
        m = re.search(r"\nbefore net:PSNR=([\d.]+)\nafter quantized net:PSNR=([\d.]+)\ntime:(\d+)us\n", r.stdout)
        pf = re.findall(r"PSNR of Frame (\d+):([\d.]+)", r.stdout)
        us = int(m.group(3))
        out[mode] = {"time_us": us, "mpixels_s_pcie_inclusive": round(n * h * w / us, 1), "psnr_before": m.group(1),
                     "psnr_after": m.group(2), "per_frame_lines": len(pf)}
        if mode.startswith("streaming"):
            # the clip is read (and written) while the GPU works: the time includes the files
            out[mode]["file_time_included"] = True
            out[mode]["report"] = next((l for l in r.stderr.splitlines() if l.startswith("stream:")), "")
            out[mode]["same_psnr_lines_as_pipelined"] = pf == out["_pf"]
        print(f"[{name}] {mode}: {us} us", file=sys.stderr, flush=True)
        if mode == "pipelined":
            # oracle on a sample of frames: the printed per-frame PSNR must match to the last decimal
            ok = True
            for i in sorted({0, distinct - 1, n // 2, n - 1})[:check]:
                rec = c_oracle.forward_blu(mb, anc[i % distinct])
                want = "%f" % c_oracle.psnr(rec, ori[i % distinct])
                ok = ok and (pf[i] == (str(i), want))
            out["per_frame_psnr_matches_oracle"] = ok
            out["frames_checked"] = min(check, 4)
            out["_pf"] = pf
    del out["_pf"]
    shutil.rmtree(d)
    return out


def run_config3(h=1080, w=1920):
    """config 3: the four QP models loaded back to back into one engine (the reference needs a rebuild per QP,
    inference/kernel.cu:122), one 1080p frame each, checked whole against the oracle; reports what a
    model switch costs (parse + integer-path proof + fragment packing + upload)."""
    from oracle import c_oracle
    from qcnn_gpu_amd.qvrcnn import qvrcnn
    out = {"height": h, "width": w, "qps": {}}
    with qvrcnn(0, 1, 1, h, w) as net:
        for rep in range(2):  # second round: every model has been loaded once, nothing is first-use any more
            for qp in (22, 27, 32, 37):
                mb = synth.synth_model(qp).to_nchw_vect_c_bytes()
                y = synth.synth_frames(h, w, qp)[1][0]
                net.synchronize()
                t0 = time.perf_counter()
                net.load_static_para(mb)
                t_load = time.perf_counter() - t0
                net.load_data(y)
                t0 = time.perf_counter()
                net.forward_blu()
                net.synchronize()
                t_fwd = time.perf_counter() - t0
                rec = net.download()
                rec_o = c_oracle.forward_blu(mb, np.ascontiguousarray(y))
                out["qps"][str(qp)] = {"model_switch_us": round(t_load * 1e6, 1), "forward_us": round(t_fwd * 1e6, 1),
                                       "bit_exact_whole_frame": bool(np.array_equal(rec, rec_o))}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scratch", default=None)
    ap.add_argument("--check", type=int, default=4)
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(dir=a.scratch)
    try:
        res = {"config3_four_qps_1080p": run_config3(),
               "config5_1080p_500f_qp27": run_case(tmp, "c5", 1080, 1920, 27, 50, 10, a.check),
               "config4_4k_64f_qp32": run_case(tmp, "c4", 2160, 3840, 32, 8, 8, a.check)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(res))
