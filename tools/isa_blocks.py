#!/usr/bin/env python3
"""Diagnostic: instruction-class histogram per basic block of one kernel in a hipcc -S listing.

    python tools/isa_blocks.py listing.s kernel_symbol [min_instructions]
"""
import collections
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])


def klass(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_barrier", "s_cbranch", "s_branch")):
        return "ctl"
    if op.startswith("s_"):
        return "salu"
    return "other"


blocks, name, cur, ops = [], "entry", collections.Counter(), collections.Counter()
for i in range(start + 1, end + 1):
    l = lines[i].strip()
    if not l or l.startswith((";", ".")) and not re.match(r"^\.LBB\d+_\d+:", l):
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((name, cur, ops))
        name, cur, ops = m.group(1), collections.Counter(), collections.Counter()
        continue
    op = l.split()[0]
    cur[klass(op)] += 1
    if klass(op) == "valu":
        ops[op] += 1
blocks.append((name, cur, ops))
for name, c, ops in blocks:
    n = sum(c.values())
    if n >= minn:
        print(f"{name:12s} total {n:5d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
        print("             " + " ".join(f"{k}:{v}" for k, v in ops.most_common(14)))
