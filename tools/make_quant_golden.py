#!/usr/bin/env python3 -B
"""Generates tests/golden/quantization_vectors.json by importing the reference's
training/quantization.py (numpy / pickle / struct only) in THIS container.

Run as:  PYTHONDONTWRITEBYTECODE=1 python3 -B tools/make_quant_golden.py
The reference does not travel to the GPU box; only the JSON produced here does.
"""
import json
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/training")
import quantization as q  # noqa: E402  (reference module, imported for fixture generation only)

out = {"mul_shift": [], "mul_shift_f": [], "quant_qfp_layer": [], "quant_qfp_concat": [], "quant_qfp_last": [],
       "adjust_quant": []}
for max_u in [1, 2, 3, 50, 127, 128, 129, 255, 1000, 3849, 4526, 4923, 5312, 6020, 6600, 6983, 7426, 7580, 8390,
              10182, 10354, 10523, 10629, 11030, 11512, 13154, 13622, 18635, 65535, 1000003, 2 ** 24 + 1]:
    out["mul_shift"].append([max_u, list(q.mul_shift(max_u))])
for ratio in [0.9, 1.0, 3.7, 17.25, 111.0, 999.5, 1170.3, 4096.0, 5000.1, 123456.7]:
    out["mul_shift_f"].append([ratio, list(q.mul_shift_f(ratio))])
for ratio, stepw, blu in [(255, 0.0077, 0.349), (255, 0.0089, 0.294), (255, 0.00736, 0.1111), (300.5, 0.01, 0.2)]:
    out["quant_qfp_layer"].append([[ratio, stepw, blu], q.quant_qfp_layer(ratio, stepw, blu)])
for args in [(365.12, 0.0087, 0.243, 0.0080, 0.243), (432.75, 0.0113, 0.172, 0.0055, 0.172),
             (523.6, 0.0067, 0.169, 0.0117, 0.169), (1146.19, 0.0127, 0.05, 0.0108, 0.05),
             (400.0, 0.01, 0.1, 0.02, 0.3)]:
    out["quant_qfp_concat"].append([list(args), q.quant_qfp_concat(*args)])
for ratio, stepw in [(754.16, 0.0025), (1261.7, 0.0012), (1016.5, 0.0016), (5790.4, 0.0017)]:
    out["quant_qfp_last"].append([[ratio, stepw], q.quant_qfp_last(ratio, stepw)])
for stepw_in, blu_in in [
    ([0.0077, 0.0087, 0.0080, 0.0067, 0.0117, 0.0025], [0.349, 0.243, 0.243, 0.169, 0.169, 0]),
    ([0.0089, 0.0113, 0.0055, 0.0107, 0.0124, 0.0012], [0.294, 0.172, 0.172, 0.101, 0.101, 0]),
    ([0.0078, 0.0075, 0.0076, 0.0108, 0.0177, 0.0016], [0.316, 0.198, 0.198, 0.125, 0.125, 0]),
    ([0.0074, 0.0127, 0.0108, 0.0030, 0.0114, 0.0017], [0.1111, 0.05, 0.05, 0.022, 0.022, 0]),
]:
    out["adjust_quant"].append([[stepw_in, blu_in], q.adjust_quant(stepw_in, blu_in)])

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "quantization_vectors.json")
with open(dst, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", os.path.normpath(dst))
