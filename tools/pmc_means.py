#!/usr/bin/env python3
"""Mean per dispatch of every counter in the rocprofv3 --pmc passes under gpurun_out/prof/, per fused kernel,
with the derived figures the design notes quote."""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof")
means = {}
for d in (sys.argv[1:] or sorted(glob.glob(P + "/pmc_*"))):  # or: the pass directories named on the command line
    fs = sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        n = r["Kernel_Name"]
        if "k_c1c2" in n or "k_c3c4" in n:
            k = "k_c1c2" if "c1c2" in n else ("k_c3c4s" if "c3c4s" in n else "k_c3c4")
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for kc, v in acc.items():
        v = sorted(v)[len(v) // 4:]          # drop the small launches of the clock ramp (4 x fewer frames)
        means[kc] = (sum(v) / len(v), len(v))
for (k, c), (m, n) in sorted(means.items()):
    print(f"{os.path.basename(k):9s} {c:28s} {n:4d} {m:14.6g}")
for k in sorted({k for k, _ in means}):
    g = lambda c: means.get((k, c), (float('nan'), 0))[0]
    cyc = g("GRBM_GUI_ACTIVE") / 8
    print(k, "kernel cycles %.4g" % cyc, "| MFMA busy %.1f %%" % (100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * cyc)),
          "| VALU(incl. MFMA)/MFMA %.2f" % (g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA")),
          "| LDS conflict %.1f %%" % (100 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")),
          "| LDS idx active / kernel cycles / 256 CU %.2f" % (g("SQ_LDS_IDX_ACTIVE") / cyc / 256))
