#!/usr/bin/env python3
"""tools/corr_energy.py <outdir>: time, package power and shader clock of k_corr4f_main<896> alone (tools/native/corrbench:
4096 channels x 65536 samples, 3000 launches back to back) for the product build and for CE_DBG builds that leave parts of
the tile loop out (tools/mkvariant.sh fdbgN aisx_lib -DCE_DBG=N, built beforehand in the container -- their results are
wrong, only their time and power count):
    f       everything                          fdbg8   no threshold test / hit path
    fdbg9   ... and no pass-through stores      fdbg10  ... and no window loads (stores kept)
    fdbg11  compute only (no loads, no stores)  fdbg88  memory only (no transform passes)
Power and clock come from the hwmon files of THIS process's GPU (bench.PowerSampler), sampled while corrbench runs.
A second argument names other tools/scratch/libaisx_<name>.so builds (comma separated; f = the product).
Writes <outdir>/corr_energy.json; tools/summarize_round.py copies it to profiles/rNN_corr_energy.json with the energy model."""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
B = os.path.join(ROOT, "tools", "native", "corrbench")
rows = []
idle = None
ps = bench.PowerSampler(0, period=0.02)
if ps.files is not None:
    ps.start()
    time.sleep(1.0)
    idle = ps.stop()
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f", "fdbg8", "fdbg9", "fdbg10", "fdbg11", "fdbg88"]
for v in names:
    lib = os.path.join(ROOT, "gr-ais_amd", "lib", "libaisx.so") if v == "f" else os.path.join(ROOT, "tools", "scratch", "libaisx_%s.so" % v)
    if not os.path.exists(lib):
        continue
    p = subprocess.Popen([B, lib, "--iters", "3000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    time.sleep(1.2)  # (warm-up: allocation, fill, clocks)
    ps = bench.PowerSampler(0, period=0.02).start()
    txt = p.communicate(timeout=180)[0]
    pw = ps.stop()
    m = re.search(r"main kernel avg ([0-9.]+) ms", txt)
    rows.append({"build": v, "kernel_ms": float(m.group(1)) if m else None, "power": pw})
    print(v, rows[-1], flush=True)
json.dump({"idle": idle, "builds": rows}, open(os.path.join(out, "corr_energy.json"), "w"), indent=1)
