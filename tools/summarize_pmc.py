#!/usr/bin/env python3
"""PMC per-launch averages of the correlator and the timing-recovery kernel from the
rocprofv3 --pmc passes under <dir> (pmc_* / msk_*) -> profiles/<tag>_corr_main_pmc.json,
profiles/<tag>_msk_sq_counters.json."""
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profile import pmc_avg  # noqa: E402

src = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
c, n = pmc_avg(glob.glob(os.path.join(src, "pmc_*")), "k_corr4_main")
m, nm = pmc_avg(glob.glob(os.path.join(src, "msk_*")), "k_msk<")
def write_msk(m):
    W, pairs = m["SQ_WAVES"], 16384
    msk = {
        "kernel": "k_msk<false,false,%d> inside the whole-flowgraph chain (about 107 time_est tags per channel and step)" % int(4096 / W),
        "command": "rocprofv3 --kernel-trace --pmc <set> --kernel-include-regex 'k_msk<' --output-format csv -- python bench.py "
                   "--steps 2 --warmup 1 --single-chain --no-cpu-baseline  (two passes)",
        "per_launch": m,
        "per_wave_and_pair_of_iterations": {k: v / W / pairs for k, v in m.items() if k != "SQ_WAVES"},
        "notes": "%d waves (workgroups of four, %d channels per wave), 16384 (even, odd) iteration pairs per channel and "
                 "launch. Cycle counters are quad-cycles per wave: x4 for cycles." % (int(W), int(4096 / W)),
    }
    json.dump(msk, open("profiles/%s_msk_sq_counters.json" % tag, "w"), indent=1)
    print(json.dumps(msk["per_wave_and_pair_of_iterations"]))


if not c:  # only the timing-recovery passes were collected this time
    write_msk(m)
    sys.exit(0)
hbm = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
tiles = 4096 * 21 / (c["SQ_WAVES"] / 4)  # tiles walked by one workgroup (4 waves)
pmc = {
    "kernel": "k_corr4_main",
    "workload": {"channels": 4096, "samples": 65536, "template_len": 896},
    "command": "rocprofv3 --kernel-trace --pmc <set> --kernel-include-regex k_corr4_main --output-format csv -- python bench.py "
               "--steps 1 --warmup 1 --chain corr --no-cpu-baseline  (one pass per counter set: FETCH_SIZE | WRITE_SIZE | "
               "SQ_INSTS_* | SQ_*CYCLES; averages over the %d launches of each pass)" % n["FETCH_SIZE"],
    "per_launch": {"FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"], "SQ_WAVES": c["SQ_WAVES"],
                   "SQ_INSTS_VALU": c["SQ_INSTS_VALU"], "SQ_INSTS_SALU": c["SQ_INSTS_SALU"], "SQ_INSTS_LDS": c["SQ_INSTS_LDS"],
                   "SQ_WAVE_CYCLES": c["SQ_WAVE_CYCLES"], "SQ_ACTIVE_INST_ANY": c["SQ_ACTIVE_INST_ANY"],
                   "SQ_WAIT_ANY": c["SQ_WAIT_ANY"], "SQ_WAIT_INST_ANY": c["SQ_WAIT_INST_ANY"]},
    "corrections": "gfx950: FETCH_SIZE counts 128-B fabric requests at 64 B => doubled (MI355X_MICROARCH.md, HBM section); "
                   "WRITE_SIZE as reported; both in KiB",
    "hbm_bytes_per_launch": hbm,
    "algorithmic_bytes_per_launch": 4294967296,
    "notes": "traffic = %.2f x algorithmic: the N-sample overlap of neighbouring tiles is served by L2. SQ cycle counters are in "
             "quad-cycles; %d waves of %.0f tiles each (3200 outputs per tile): %.0f VALU + %.0f SALU + %.0f LDS instructions per wave "
             "and tile; a wave is issuing %.0f %% of its lifetime, three waves per SIMD."
             % (hbm / 4294967296, int(c["SQ_WAVES"]), tiles, c["SQ_INSTS_VALU"] / c["SQ_WAVES"] / tiles,
                c["SQ_INSTS_SALU"] / c["SQ_WAVES"] / tiles, c["SQ_INSTS_LDS"] / c["SQ_WAVES"] / tiles,
                100 * c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]),
}
json.dump(pmc, open("profiles/%s_corr_main_pmc.json" % tag, "w"), indent=1)
print(pmc["notes"])
if m:
    write_msk(m)
