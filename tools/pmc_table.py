#!/usr/bin/env python3
"""Per-kernel averages of every counter found under a tools/pmc_passes.sh output directory:
   python tools/pmc_table.py <dir> [kernel substring]"""
import csv
import glob
import json
import os
import sys


def table(src, sub=""):
    acc = {}
    for f in glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    dur = {}
    for f in glob.glob(os.path.join(src, "p0", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                dur.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    out = {}
    for k, v in acc.items():
        out[k] = {c: sum(x) / len(x) for c, x in v.items()}
        out[k]["launches"] = max(len(x) for x in v.values())
        if k in dur:
            out[k]["avg_ms_profiled"] = sum(dur[k]) / len(dur[k])
    return out


if __name__ == "__main__":
    t = table(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    print(json.dumps(t, indent=1, sort_keys=True))
