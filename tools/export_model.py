#!/usr/bin/env python3
"""Float checkpoint dump -> the files the inference driver and the training side load.

    python tools/export_model.py vrcnn_16.0_qfp_37.data 37 [outdir]

Input: the file VRCNN.dump writes (training/model.py:325-337).  Output in outdir (default "."):
    quant_params{QP}.data, quant_params_cpp_{QP}.data     parameter rows (training/quantization.py:89-96)
    qvrcnn_hwcn_8bit_qfp_{QP}.data                        int8 model, HWCN (input of convert_model)
    qvrcnn_nchw_vect_c_8bit_qfp_{QP}.data                 int8 model, NCHW_VECT_C (what QCNN_GPU loads)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcnn_gpu_amd import quant  # noqa: E402


def main():
    if len(sys.argv) < 3:
        print(__doc__)
        return 1
    src, qp = sys.argv[1], int(sys.argv[2])
    out = sys.argv[3] if len(sys.argv) > 3 else "."
    if qp not in quant.BLU_INIT:
        print(f"no initial thresholds for QP {qp}; known: {sorted(quant.BLU_INIT)}")
        return 1
    with open(src, "rb") as f:
        rows, model = quant.export(f.read(), qp)
    os.makedirs(out, exist_ok=True)
    files = {
        f"quant_params{qp}.data": quant.quant_params_pickle(rows),
        f"quant_params_cpp_{qp}.data": quant.quant_params_cpp(rows),
        f"qvrcnn_hwcn_8bit_qfp_{qp}.data": model.to_hwcn_bytes(),
        f"qvrcnn_nchw_vect_c_8bit_qfp_{qp}.data": model.to_nchw_vect_c_bytes(),
    }
    for name, data in files.items():
        with open(os.path.join(out, name), "wb") as f:
            f.write(data)
        print(f"wrote {os.path.join(out, name)} ({len(data)} bytes)")
    for name, row in zip(("C1", "C2_1", "C2_2", "C3_1", "C3_2", "C4"), rows):
        print(name, row)
    return 0


if __name__ == "__main__":
    sys.exit(main())
