#!/usr/bin/env python3
"""Instruction mix of a line range of a gfx950 .s file: python tools/isa_count.py file.s lo hi"""
import collections
import re
import sys

def classify(m):
    if m.startswith("v_pk_"):
        return "valu_pk"
    if m.startswith("v_"):
        return "valu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if m.startswith("s_waitcnt"):
        return "waitcnt"
    if m.startswith("s_barrier"):
        return "barrier"
    if m.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if m.startswith("s_load"):
        return "smem"
    if m.startswith("s_"):
        return "salu"
    return "other"

def count(path, lo, hi, detail=False):
    cat = collections.Counter()
    ops = collections.Counter()
    for i, line in enumerate(open(path), 1):
        if i < lo or i > hi:
            continue
        s = line.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":") or re.match(r"^[.\w]+:", s):
            continue
        m = s.split()[0]
        cat[classify(m)] += 1
        ops[m] += 1
    return cat, ops

if __name__ == "__main__":
    cat, ops = count(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
    print(dict(cat), "total", sum(cat.values()))
    if len(sys.argv) > 4:
        for k, v in ops.most_common(60):
            print("  %-28s %d" % (k, v))
