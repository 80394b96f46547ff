import os, subprocess, sys, numpy as np
sys.path.insert(0, os.getcwd())
from qcnn_gpu_amd import formats, synth
import tempfile, pathlib
tmp = pathlib.Path(tempfile.mkdtemp())
for (h, w, nf) in ((72, 120, 37), (1080, 1920, 40)):
    qp = 27
    ori, anc = synth.synth_frames(h, w, qp, frames=min(nf, 4), seed=91)
    ori = np.concatenate([ori] * (nf // len(ori) + 1))[:nf]; anc = np.concatenate([anc] * (nf // len(anc) + 1))[:nf]
    formats.write_yuv420_luma(str(tmp / "seq.yuv"), ori); formats.write_yuv420_luma(str(tmp / f"anchor_Q{qp}.yuv"), anc)
    (tmp / f"model_{qp}.data").write_bytes(synth.synth_model(qp).to_nchw_vect_c_bytes())
    for mode in ("QCNN_STREAM", "QCNN_PIPELINE", "NONE"):
        e = dict(os.environ, QCNN_MODEL=str(tmp / "model_%d.data"), QCNN_QPS=str(qp), QCNN_FRAMES=str(nf)); e[mode] = "1"; e["QCNN_REPORT_RSS"] = "1"
        r = subprocess.run(["qcnn_gpu_amd/bin/QCNN_GPU", str(tmp / "seq.yuv"), str(tmp / "anchor_"), str(h), str(w)], cwd=os.getcwd(), env=e, capture_output=True, text=True)
        print(h, w, nf, mode, [l.strip() for l in r.stderr.splitlines() if "max RSS" in l or "stream:" in l])
