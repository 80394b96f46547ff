#!/usr/bin/env python3
"""Turns what tools/profile_r02.sh (round 1: tools/profile_gpu.sh) left under gpurun_out/prof/ into the committed
summaries under profiles/ (kernel statistics, counter means per dispatch, traffic.json).  Picks the newest file
of every pass."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    if not files:
        raise SystemExit(f"nothing matches {pattern}")
    return files[-1]


shutil.copy(newest(P + "/stats/vm/*_kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(P, "bench_stats.json"), os.path.join(ROOT, "profiles", f"{tag}_bench_under_rocprof.json"))
rows, means = [], {}
for p in ("sq", "lds", "act", "fetch", "write"):
    if not glob.glob(f"{P}/pmc_{p}/vm/*_counter_collection.csv"):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(f"{P}/pmc_{p}/vm/*_counter_collection.csv"))):
        if "k_c1c2" in r["Kernel_Name"] or "k_c3c4" in r["Kernel_Name"]:
            k = "k_c1c2" if "c1c2" in r["Kernel_Name"] else "k_c3c4"   # k_c3c4: kernel B, whichever form ran
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        rows.append((p, k, c, len(v), sum(v) / len(v)))
        means[(k, c)] = sum(v) / len(v)
with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_counters.csv"), "w") as f:
    f.write("# rocprofv3 --pmc passes of a short `python3 bench.py` run (16 frames of 1920x1080 per dispatch; see the profiling script); mean per dispatch\n")
    f.write("pass,kernel,counter,dispatches,mean_value\n")
    for r in rows:
        f.write("%s,%s,%s,%d,%.6g\n" % r)
fpl = 16
a = means[("k_c1c2", "FETCH_SIZE")] * 1024 + means[("k_c1c2", "WRITE_SIZE")] * 1024
b = 2 * means[("k_c3c4", "FETCH_SIZE")] * 1024 + means[("k_c3c4", "WRITE_SIZE")] * 1024
json.dump({
    "height": 1080, "width": 1920, "frames_per_launch": fpl,
    "k_c1c2_hbm_bytes_per_launch": round(a), "k_c3c4_hbm_bytes_per_launch": round(b),
    "raw_counters_KiB_per_launch": {f"{k}_{c}": v for (k, c), v in means.items() if c in ("FETCH_SIZE", "WRITE_SIZE")},
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_gpu.sh), KiB per dispatch "
              "averaged over the dispatches of a short bench run, 16 frames per dispatch. k_c3c4 reads conc1 with 16-byte-per-lane "
              "loads, for which FETCH_SIZE on gfx950 reports half the bytes: doubled. k_c1c2 reads Y with byte-wide loads "
              "(uncalibrated width): taken as reported. Byte-wide stores of k_c3c4 (x_rec) are uncalibrated and read 2x the frame bytes.",
    "algorithmic_bytes_per_frame": {"k_c1c2": 1080 * 1920 * (1 + 48), "k_c3c4": 1080 * 1920 * (48 + 1 + 1)},
}, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
cyc = {k: means[(k, "GRBM_GUI_ACTIVE")] / 8 for k in ("k_c1c2", "k_c3c4")}
for k in ("k_c1c2", "k_c3c4"):
    print(k, "kernel cycles %.3g" % cyc[k], "MFMA busy %.1f %%" % (100 * means[(k, "SQ_VALU_MFMA_BUSY_CYCLES")] / (1024 * cyc[k])),
          "VALU/MFMA %.2f" % (means[(k, "SQ_INSTS_VALU")] / means[(k, "SQ_INSTS_MFMA")]),
          "LDS conflict %.1f %%" % (100 * means[(k, "SQ_LDS_BANK_CONFLICT")] / means[(k, "SQ_LDS_IDX_ACTIVE")]),
          "bytes/frame %.1f MB" % ((a if k == "k_c1c2" else b) / fpl / 1e6))
print(open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv")).read())
