#!/usr/bin/env python3
"""profiles/r03_corr_main_pmc.json and profiles/r03_chain_traffic.json from one gpurun_out/<dir>
that holds  corr_pmc_table.json (tools/pmc_table.py over tools/pmc_passes.sh passes of
tools/native/corrbench) and chain_FETCH_SIZE/, chain_WRITE_SIZE/ (rocprofv3 --pmc passes of a short
whole-flowgraph run):   python tools/summarize_r03.py gpurun_out/r3c12 r03"""
import csv
import glob
import json
import os
import statistics
import sys

src, tag = sys.argv[1], sys.argv[2]
t = json.load(open(os.path.join(src, "corr_pmc_table.json")))
name = [k for k in t if k.startswith("k_corr4d_main")][0]
c = t[name]
hbm = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
W = c["SQ_WAVES"]
tiles_per_wave = 4096 * 21 / (W / 4)
per = lambda k: c[k] / W / tiles_per_wave
act, wait_any, stall = c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], None
stall = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
out = {
    "kernel": name,
    "workload": {"channels": 4096, "samples": 65536, "template_len": 896},
    "command": "tools/pmc_passes.sh <dir> k_corr4 -- tools/native/corrbench gr-ais_amd/lib/libaisx.so --iters 5   (rocprofv3 "
               "--kernel-trace --pmc <set> --kernel-include-regex k_corr4 --output-format csv, one pass per counter set: SQ cycles | "
               "SQ instructions | FETCH_SIZE | WRITE_SIZE | LDS | TCC; averages over the launches of each pass; tools/pmc_table.py, "
               "tools/summarize_r03.py)",
    "per_launch": {k: v for k, v in c.items() if k != "launches"},
    "corrections": "gfx950: FETCH_SIZE counts 128-B fabric requests at 64 B => doubled (MI355X_MICROARCH.md, HBM section; calibrated "
                   "on the kernel's own access shapes in profiles/r02_traffic_calibration.json); WRITE_SIZE as reported; both KiB",
    "hbm_bytes_per_launch": hbm,
    "algorithmic_bytes_per_launch": 4294967296,
    "traffic_over_algorithmic": hbm / 4294967296,
    "occupancy": {"workgroups_per_CU": 2, "waves_per_SIMD": 2, "VGPRs_per_lane": 256, "LDS_bytes_per_workgroup": 71680,
                  "waves_per_launch": W, "tiles_per_wave": tiles_per_wave,
                  "why": "256 VGPRs and 71 680 B of LDS per workgroup both allow exactly two workgroups of four waves per CU"},
    "per_wave_and_tile": {"VALU": per("SQ_INSTS_VALU"), "SALU": per("SQ_INSTS_SALU"), "LDS": per("SQ_INSTS_LDS"),
                          "VMEM_RD": per("SQ_INSTS_VMEM_RD"), "VMEM_WR": per("SQ_INSTS_VMEM_WR"), "BRANCH": per("SQ_INSTS_BRANCH")},
    "wave_lifetime_fractions": {"executing_an_instruction": act, "parked_at_s_waitcnt_or_barrier": wait_any,
                                "stalled_at_issue": stall,
                                "counters": "SQ_ACTIVE_INST_ANY, SQ_WAIT_ANY, SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES"},
    "lds": {"busy_fraction_of_kernel_cycles": c["SQ_LDS_IDX_ACTIVE"] / 4.0 / c["GRBM_GUI_ACTIVE"] / 256.0 * 4.0 if False else None,
            "bank_conflict_fraction_of_lds_cycles": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]},
    "instruction_budget": {
        "text": "per launch 86 016 tiles x 4 waves; a wave and tile costs %.0f VALU (about 690 of them packed butterflies at ~2.2 ns of "
                "their SIMD each, the rest ~1.1 ns: tools/ubench/valu_tput.hip) => %.2f ms of pure VALU time on 1024 SIMDs; "
                "HBM at the 5.1 TB/s the best plain copy reaches on this chip (bench.py: copy_ceiling_GBs) needs %.2f ms for the "
                "4.29 GB; the launch takes %.2f ms under the profiler (1.15-1.16 ms without): VALU, LDS and memory phases of a tile "
                "overlap only as far as two waves per SIMD allow" % (
                    per("SQ_INSTS_VALU"), 86016 * 4 * (690 * 2.2 + (per("SQ_INSTS_VALU") - 690) * 1.1) / 1024 / 1e6, 4.294967296 / 5.1, c["avg_ms_profiled"]),
    },
    "notes": "round 3: the second LDS exchange of each direction is a wave-level one (writer and reader share t >> 4): three workgroup "
             "barriers per tile instead of five; 1.148-1.158 ms against 1.172 per launch (corr-only run, hipEvents, one box).  "
             "Variants measured and not kept: W_256 twiddles read two at a time (spills at 256 VGPRs: 1.40 ms), max-norm pre-test of the "
             "threshold (13 VALU fewer per wave and tile: 1.150 against 1.149 ms).  SQ cycle counters are quad-cycles per wave.",
}
del out["lds"]["busy_fraction_of_kernel_cycles"]
json.dump(out, open("profiles/%s_corr_main_pmc.json" % tag, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("traffic_over_algorithmic", "wave_lifetime_fractions")}))

# chain traffic
ker = {}
for which, key in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    for f in glob.glob(os.path.join(src, "chain_" + which, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != which:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            ker.setdefault(k, {}).setdefault(key, []).append(float(r["Counter_Value"]))
rows = {}
for k, v in ker.items():
    if not k.startswith("k_"):
        continue
    f = statistics.median(v.get("fetch", [0.0])) * 2 * 1024 / 1e9
    w = statistics.median(v.get("write", [0.0])) * 1024 / 1e9
    rows[k] = {"launches": len(v.get("fetch", [])), "fetch_GB": f, "write_GB": w}
stream = [k for k in rows if k.startswith(("k_fs_est", "k_fs_walk", "k_agc8", "k_corr4d_main", "k_corr_resolve"))]
chain = {
    "what": "HBM bytes per launch of every kernel of the default whole-flowgraph step (4096 channels x 65536 samples, N = 896, pipelined "
            "chain aisx_chain_step): rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes, medians over the launches "
            "of a 5-step run",
    "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python bench.py --steps 3 --warmup 2 "
               "--single-chain --no-cpu-baseline --parity-channels 0",
    "corrections": "FETCH_SIZE doubled (gfx950: 128-byte requests counted as 64; calibrated in r02_traffic_calibration.json), WRITE_SIZE as reported, both KiB",
    "kernels": rows,
    "streaming_side_GB_per_step": sum(rows[k]["fetch_GB"] + rows[k]["write_GB"] for k in stream),
    "whole_step_GB": sum(v["fetch_GB"] + v["write_GB"] for v in rows.values()),
    "notes": "unchanged from round 2 within 1 %: the round's changes are scheduling (the chain as a product API, the head start of the "
             "recovery kernel) and the correlator's barriers, none of which moves bytes.  What would: the AGC folded into the "
             "correlator's window load (-4.3 GB), the delayed pass-through read by the timing recovery as a shifted view of the "
             "correlator's input (-2.15 GB) -- DESIGN.md section 9.",
}
json.dump(chain, open("profiles/%s_chain_traffic.json" % tag, "w"), indent=1)
print("streaming side %.2f GB, whole step %.2f GB" % (chain["streaming_side_GB_per_step"], chain["whole_step_GB"]))
