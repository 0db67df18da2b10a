#!/usr/bin/env python3
"""Per-kernel launch count / mean / min duration from a rocprofv3 --kernel-trace csv directory."""
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].replace("void ", "").split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = v[len(v) // 4:]  # skip warm-up
    print("%-40s n=%4d mean %.3f ms  min %.3f  (after warm-up: mean %.3f)" % (k[:40], len(v), sum(v) / len(v), min(v), sum(v2) / len(v2)))
