#!/usr/bin/env python3
"""profiles/<tag>_* from one tools/profile_round.sh output directory:
    python tools/summarize_round.py gpurun_out/prof_r04 r04
  <tag>_default_bench_line.json          bench.py's line, plain run (no profiler)
  <tag>_default_bench_kernel_stats.{md,csv}  rocprofv3 --kernel-trace --stats of the same command (+ that run's line)
  <tag>_corr_main_pmc.json, <tag>_corr2d_main_pmc.json   PMC passes of the two correlator builds alone
  <tag>_chain_traffic.json               HBM bytes per kernel and step (FETCH_SIZE / WRITE_SIZE passes)
  <tag>_msk_sq_counters.json             the timing-recovery kernel inside the chain (SQ counters)
  <tag>_msk_time_parallel.json           the opt-in time-parallel recovery: restart points, units, junctions, step time"""
import csv
import glob
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import summarize_profile as sp  # noqa: E402

src, tag = sys.argv[1], sys.argv[2]


def bench_line(path):
    line = None
    for ln in open(path):
        if ln.startswith("{"):
            line = json.loads(ln)
    return line


plain = bench_line(os.path.join(src, "bench_default.log"))
prof = bench_line(os.path.join(src, "bench_profiled.log"))
json.dump(plain, open("profiles/%s_default_bench_line.json" % tag, "w"), indent=1)

# ---- kernel stats of the profiled default run
# (rocprofv3 follows the child processes bench.py starts -- the HBM-ceiling micro-benchmark, the GNU Radio harness -- and
# writes a trace per process: the benchmark's own is the one that holds the timing-recovery kernel)
def _pick(pattern):
    cands = glob.glob(os.path.join(src, "stats", "*", pattern))
    best = max(cands, key=lambda f: open(f).read().count("k_msk<"))
    return best


trace = _pick("*kernel_trace.csv")
stats = trace.replace("kernel_trace.csv", "kernel_stats.csv")
st = {k: v for k, v in sp.timed_stats(trace, prof["steps"], prof["warmup"]).items()
      if not k.startswith(("k_mskp", "k_msk_ff"))}  # (kernels of the time-parallel side run only)
open("profiles/%s_default_bench_kernel_stats.csv" % tag, "w").write(open(stats).read())
md = ["# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (MI355X, default run)", "",
      "Command on the GPU box: `bash tools/profile_round.sh %s` (`cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 "
      "--kernel-trace --stats --output-format csv -d gpurun_out/prof_%s/stats -- python bench.py --no-cpu-baseline`); summarised by "
      "`tools/summarize_round.py`." % (tag, tag), "",
      "bench.py's line of the SAME (profiled) run: %.0f complex MS/s, %.2f ms/step, `roofline.kernel_ms` %.3f (hipEvents over the "
      "timed region), `kernel_ms_alone` %.3f.  The plain run right before it on the same box (`%s_default_bench_line.json`): "
      "%.0f complex MS/s, %.2f ms/step, `roofline.kernel_ms` %.3f."
      % (prof["value"], prof["ms_per_step"], prof["roofline"]["kernel_ms"], prof["roofline"]["kernel_ms_alone"], tag, plain["value"],
         plain["ms_per_step"], plain["roofline"]["kernel_ms"]), "",
      "Per kernel, the %d TIMED launches of the whole-flowgraph chain only (the run also holds %d warm-up steps, the isolated "
      "correlator launches, the corr_est -> msk-only chain, the no-look-ahead and the time-parallel side runs; rocprofv3's own `--stats` "
      "table over ALL launches is in `%s_default_bench_kernel_stats.csv`):" % (prof["steps"], prof["warmup"], tag), "",
      "| kernel | timed launches | avg ms | min ms | max ms | launches in the whole run |", "|---|---|---|---|---|---|"]
for k, v in sorted(st.items(), key=lambda kv: -kv[1]["avg"]):
    md.append("| %s | %d | %.3f | %.3f | %.3f | %d |" % (k[:48], v["n"], v["avg"], v["mn"], v["mx"], v["calls"]))
kname = prof["roofline"]["kernel"]
for k in st:
    if k == kname or k.startswith(kname + "<"):
        md += ["", "`%s`: %.3f ms here against `roofline.kernel_ms` = %.3f ms in the same run's bench line." % (
            k, st[k]["avg"], prof["roofline"]["kernel_ms"])]
open("profiles/%s_default_bench_kernel_stats.md" % tag, "w").write("\n".join(md) + "\n")
print("\n".join(md[-12:]))


# ---- the correlator builds alone
def corr_summary(table_json, prefix, N, tiles_per_chan, waves_per_wg, lds_bytes, out_name, notes, vgprs=256):
    t = json.load(open(table_json))
    name = [k for k in t if k.startswith(prefix)][0]
    c = t[name]
    hbm = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    W = c["SQ_WAVES"]
    alg = 16.0 * 4096 * 65536
    tiles_per_wave = 4096 * tiles_per_chan / (W / waves_per_wg)
    per = lambda k: c[k] / W / tiles_per_wave  # noqa: E731
    out = {
        "kernel": name,
        "workload": {"channels": 4096, "samples": 65536, "template_len": N},
        "command": "tools/pmc_passes.sh <dir> %s -- tools/native/corrbench gr-ais_amd/lib/libaisx.so%s --iters 5   (rocprofv3 "
                   "--kernel-trace --pmc <set> --kernel-include-regex %s --output-format csv, one pass per counter set: SQ cycles | "
                   "SQ instructions | FETCH_SIZE | WRITE_SIZE | LDS | TCC; averages over the launches of each pass; tools/pmc_table.py, "
                   "tools/summarize_round.py)" % (prefix[:8], "" if N == 896 else " --N %d" % N, prefix[:8]),
        "per_launch": {k: v for k, v in c.items() if k != "launches"},
        "corrections": "gfx950: FETCH_SIZE counts 128-B fabric requests at 64 B => doubled (MI355X_MICROARCH.md, HBM section; calibrated "
                       "on the kernel's own access shapes in profiles/r02_traffic_calibration.json); WRITE_SIZE as reported; both KiB",
        "hbm_bytes_per_launch": hbm,
        "algorithmic_bytes_per_launch": alg,
        "traffic_over_algorithmic": hbm / alg,
        "occupancy": {"waves_per_launch": W, "tiles_per_wave": tiles_per_wave, "VGPRs_per_lane": vgprs,
                      "LDS_bytes_per_workgroup": lds_bytes, "waves_per_workgroup": waves_per_wg},
        "per_wave_and_tile": {"VALU": per("SQ_INSTS_VALU"), "SALU": per("SQ_INSTS_SALU"), "LDS": per("SQ_INSTS_LDS"),
                              "VMEM_RD": per("SQ_INSTS_VMEM_RD"), "VMEM_WR": per("SQ_INSTS_VMEM_WR"), "BRANCH": per("SQ_INSTS_BRANCH")},
        "wave_lifetime_fractions": {"executing_an_instruction": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                                    "parked_at_s_waitcnt_or_barrier": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                                    "stalled_at_issue": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                                    "counters": "SQ_ACTIVE_INST_ANY, SQ_WAIT_ANY, SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES"},
        "lds_bank_conflict_fraction_of_lds_cycles": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
        "achieved_GBs_profiled": alg / (c["avg_ms_profiled"] * 1e-3) / 1e9,
        "notes": notes,
    }
    json.dump(out, open("profiles/%s" % out_name, "w"), indent=1)
    print(out_name, "traffic/algorithmic %.3f, %.3f ms under the profiler" % (out["traffic_over_algorithmic"], c["avg_ms_profiled"]))


L896 = 4096 - 896  # outputs per tile of the F = 4096 build
L112 = 2048 - 112
corr_summary(os.path.join(src, "corr896_pmc_table.json"), "k_corr4f_main", 896, -(-65536 // L896), 8, 55232,
             "%s_corr_main_pmc.json" % tag,
             "round 6: k_corr4f_main (k_corr4f.h) -- 512 threads x 8 points (8 x 8 x 8 x 8), 120 VGPRs, 55 232 B of LDS: two workgroups "
             "of eight waves per CU = four waves per SIMD (round 5's k_corr4d_main: 256 VGPRs, 71 680 B, two waves per SIMD); the next "
             "window's new items fetched into registers (raw buffer loads, nt), one window image + a double-buffered overlap buffer, "
             "two barriers per tile.  Alone on the chip the kernel runs the package into its 1400 W limit (%s_corr_energy.json): its "
             "time follows the shader clock the firmware leaves it.  SQ cycle counters are quad-cycles per wave." % tag, vgprs=120)
corr_summary(os.path.join(src, "corr112_pmc_table.json"), "k_corr2d_main", 112, -(-65536 // L112), 2, 35840,
             "%s_corr2d_main_pmc.json" % tag,
             "round 4: nt policy as in the F = 4096 build; the per-thread output mask is rebuilt where a hit needs it instead of "
             "living across the tile loop -- the N = 112 build no longer spills (255 VGPRs, no scratch).  Round 6: built without the "
             "SI load / store optimiser (no ds_read2 merging), otherwise unchanged.")

# ---- chain traffic
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
    rows[k] = {"launches": len(v.get("fetch", [])), "fetch_GB": statistics.median(v.get("fetch", [0.0])) * 2 * 1024 / 1e9,
               "write_GB": statistics.median(v.get("write", [0.0])) * 1024 / 1e9}
stream = [k for k in rows if k.startswith(("k_fs_est", "k_fs_walk", "k_agc8", "k_agcw", "k_corr4d_main", "k_corr4f_main", "k_corr_resolve"))]
prev = None
try:
    prev = json.load(open("profiles/r05_chain_traffic.json"))
except OSError:
    pass
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
    "round_5": None if prev is None else {"streaming_side_GB_per_step": prev["streaming_side_GB_per_step"], "whole_step_GB": prev["whole_step_GB"]},
    "notes": "round 6: the same passes and the same bytes as round 5 (the correlator build changed, not what it moves).  What could still "
             "be removed is unchanged: the AGC folded into the correlator's window load (-4.3 GB), the delayed pass-through as a view "
             "(-2.15 GB, measured in round 4: no gain) -- DESIGN.md says why neither was built.",
}
json.dump(chain, open("profiles/%s_chain_traffic.json" % tag, "w"), indent=1)
print("streaming side %.2f GB, whole step %.2f GB" % (chain["streaming_side_GB_per_step"], chain["whole_step_GB"]))

# ---- timing recovery: SQ counters in the chain
t = json.load(open(os.path.join(src, "msk_pmc_table.json")))
name = [k for k in t if k.startswith("k_msk<")][0]
c = t[name]
pairs = 512 * 16384.0  # waves x pairs of iterations per step (65536 samples at 4 per symbol, two iterations per symbol)
msk = {
    "kernel": name + " inside the whole-flowgraph chain (serial kernel, the default)",
    "command": "rocprofv3 --kernel-trace --pmc <set> --kernel-include-regex 'k_msk<' --output-format csv -- python bench.py --steps 4 "
               "--warmup 2 --single-chain --no-cpu-baseline --parity-channels 0   (two passes: SQ cycles | SQ instructions)",
    "per_launch": c,
    "per_wave_and_pair_of_iterations": {k: c[k] / pairs for k in c if k.startswith("SQ_") and k != "SQ_WAVES"},
    "notes": "512 waves of 8 channels (one per SIMD on 128 CUs), 16384 pairs of iterations per channel and step.  SQ cycle counters are "
             "quad-cycles per wave.  The recurrence issues one instruction every ~7 cycles of its SIMD: a dependent chain (tap row "
             "from mu -> 8-tap FIR in the reference's summation order -> error -> mu, omega), see DESIGN.md.",
}
json.dump(msk, open("profiles/%s_msk_sq_counters.json" % tag, "w"), indent=1)

# ---- the time-parallel recovery (opt-in): what the default bench run measured
tp = plain.get("msk_time_parallel")
if tp:
    rs = tp["restart_stats_last_step"]
    doc = {
        "what": "aisx_msk_set_time_parallel (k_mskp.h): restart points at pairs of time_est tags one symbol apart, units run from them "
                "on their own stream, the serial kernel as the join teleports through units whose junctions check bit for bit; "
                "set_max_noutput_items(256) bounds a stale tag's blocking to one general_work call.  OFF by default.",
        "from": "profiles/%s_default_bench_line.json: msk_time_parallel (same box, same run as the headline)" % tag,
        "ms_per_step_time_parallel": tp["ms_per_step"],
        "ms_per_step_serial_kernel_same_max_noutput_items": tp["ms_per_step_serial_kernel_same_max_noutput_items"],
        "ms_per_step_default": plain["ms_per_step"],
        "corr_kernel_ms_in_that_chain": tp["corr_kernel_ms"],
        "restart_stats_last_step": rs,
        "per_channel": {"restart_points": rs["restart_points"] / 4096.0, "units_taken": rs["units_taken"] / 4096.0,
                        "symbols_from_units_fraction": rs["symbols_from_units"] / (4096.0 * 16384.0)},
        "junctions": {"links_between_consecutive_units": rs["links"], "equal_bit_for_bit": rs["links_equal"],
                      "failed": rs["links"] - rs["links_equal"],
                      "units_thrown_away": rs["restart_points"] - rs["units_taken"],
                      "how_a_failure_is_handled": "the join compares (last interpolated sample, last squared-difference term) with what the "
                                                  "unit assumed, bitwise; a unit whose junction differs -- or that would cross a general_work "
                                                  "boundary it may not cross -- is not taken and its stretch is run by the serial loop "
                                                  "(tests/test_gpu_mskp.py::test_time_parallel_with_failing_junctions, tests/test_emul_mskp.py)"},
        "parity": tp.get("parity"),
        "why_not_default": "the critical path is the longest stretch without a restart point (about 22 000 items = 10 empty slots, a third "
                           "of a step) run at the engine's speed, the units do the same total work as the serial pass on an engine that "
                           "needs ~175 instructions per pair of iterations against ~100, and units + join share the SIMDs with the sample "
                           "passes of the next step: DESIGN.md section on the timing recovery",
    }
    json.dump(doc, open("profiles/%s_msk_time_parallel.json" % tag, "w"), indent=1)
    print("time-parallel: %.2f ms/step against %.2f (serial kernel, same Q) and %.2f (default)" % (
        tp["ms_per_step"], tp["ms_per_step_serial_kernel_same_max_noutput_items"], plain["ms_per_step"]))


# ---- round 5: the front-end kernels alone (streaming kernel against the tile kernel it replaced)
try:
    fr = {}
    for name, path in (("k_agcw", "agcw_pmc_table.json"), ("k_fs_est", "est_pmc_table.json"), ("k_agc8", "agc8_pmc_table.json")):
        if not os.path.exists(os.path.join(src, path)):
            continue
        t = json.load(open(os.path.join(src, path)))
        for k, c in t.items():
            if not k.startswith(name):
                continue
            waves = c.get("SQ_WAVES", 0.0)
            fr[k] = {"per_launch": c,
                     "per_wave": {m: c[m] / waves for m in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if m in c and waves},
                     "wave_active_fraction": c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None,
                     "wave_waiting_fraction": c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None}
    alone = {}
    for label, d in (("streaming", "front_stats"), ("tile", "front_tile_stats")):
        for f in glob.glob(os.path.join(src, d, "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Name"].split("(")[0].replace("void ", "")
                if k.startswith(("k_agcw", "k_agc8", "k_fs_est", "k_fs_walk")):
                    alone.setdefault(label, {})[k] = float(r["AverageNs"]) / 1e6
    doc = {"what": "the front-end kernels of the stock chain with the chip to themselves (tools/front_alone.py: 4096 channels x 65536 samples, "
                   "eight calls of aisx_freqsync_agc_process): SQ counters of the streaming kernel k_agcw<true> (round 5) and of the tile "
                   "kernel k_agc8 it replaced (AISX_AGC_STREAMING=0), and of k_fs_est with the single-precision prefilter of its peak search",
           "command": "rocprofv3 --kernel-trace --pmc <set> --kernel-include-regex <k> --output-format csv -- python tools/front_alone.py "
                      "(two passes: SQ cycles | SQ instructions); durations from rocprofv3 --kernel-trace --stats of the same script",
           "kernels": fr, "avg_ms_alone": alone,
           "notes": "k_agcw: a wave = one run of sixteen 512-item blocks (+ one read ahead), four waves per workgroup, no barrier after the "
                    "sine table is staged; k_agc8: 1024-thread tiles of 8192 items, eight barriers.  SQ cycle counters are quad-cycles per wave."}
    json.dump(doc, open("profiles/%s_front_end_kernels.json" % tag, "w"), indent=1)
    print("front end alone (ms):", alone)
except (OSError, KeyError, ValueError) as e:
    print("front-end kernel summary skipped:", e)
try:
    hc = json.load(open(os.path.join(src, "hbm_ceiling.json")))
    json.dump(hc, open("profiles/%s_hbm_ceilings.json" % tag, "w"), indent=1)
except (OSError, ValueError) as e:
    print("hbm ceilings skipped:", e)


# ---- round 6: config 4's per-GPU shape under the kernel trace
try:
    c4 = bench_line(os.path.join(src, "c4_profiled.log"))
    cands = glob.glob(os.path.join(src, "c4stats", "*", "*kernel_trace.csv"))
    tr = max(cands, key=lambda f: open(f).read().count("k_msk<"))
    st4 = {k: v for k, v in sp.timed_stats(tr, c4["steps"], c4["warmup"]).items() if not k.startswith(("k_mskp", "k_msk_ff"))}
    md = ["# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --config4 --steps 30", "",
          "BASELINE config 4's per-GPU shape (8192 channels x 65536 samples per step) on one MI355X.  bench.py's line of this (profiled) "
          "run: %.0f complex MS/s, %.2f ms/step, correlator %.3f ms in the chain." % (c4["value"], c4["ms_per_step"], c4["roofline"]["kernel_ms"]), "",
          "| kernel | timed launches | avg ms | min ms | max ms |", "|---|---|---|---|---|"]
    for k, v in sorted(st4.items(), key=lambda kv: -kv[1]["avg"]):
        md.append("| %s | %d | %.3f | %.3f | %.3f |" % (k[:48], v["n"], v["avg"], v["mn"], v["mx"]))
    md += ["", "The sample passes (k_fs_est -> k_agcw -> k_corr4f_main -> k_corr_resolve, one behind the other on one stream) add up to the "
           "step; the recovery (k_msk) and the phase walk (k_fs_walk) run beside them on their own streams (DESIGN.md 5)."]
    open("profiles/%s_config4_kernel_stats.md" % tag, "w").write("\n".join(md) + "\n")
    print("\n".join(md[-14:]))
except (OSError, KeyError, ValueError, TypeError) as e:
    print("config 4 summary skipped:", e)

# ---- round 6: the correlator's energy split
try:
    ce = json.load(open(os.path.join(src, "energy", "corr_energy.json")))
    idle_w = (ce.get("idle") or {}).get("package_W_mean")
    rows_e = []
    for b in ce["builds"]:
        pw = b.get("power") or {}
        w, ms = pw.get("package_W_mean"), b.get("kernel_ms")
        rows_e.append({"build": b["build"], "kernel_ms": ms, "package_W": w, "sclk_MHz": pw.get("sclk_MHz_mean"),
                       "joule_per_launch": None if (w is None or ms is None) else w * ms * 1e-3})
    what = {"f": "the product kernel", "fdbg8": "no threshold test / hit path", "fdbg9": "no hit path, no pass-through stores",
            "fdbg10": "no hit path, no window loads (stores kept)", "fdbg11": "compute only (no loads, no stores, no hit path)",
            "fdbg88": "memory only (loads, stores, overlap buffer; no transform passes)"}
    for r in rows_e:
        r["what"] = what.get(r["build"])
    doc = {"what": "k_corr4f_main<896> alone (tools/native/corrbench, 4096 channels x 65536 samples, 3000 launches back to back): kernel "
                   "time (the library's hipEvents), package power and shader clock (hwmon of the process's GPU, 20 ms samples) of the product "
                   "build and of CE_DBG builds that leave parts of the tile loop out (timing / power only: their results are wrong)",
           "command": "python tools/corr_energy.py <dir>  (variants: tools/mkvariant.sh fdbgN aisx_lib -DCE_DBG=N)",
           "package_limit_W": 1400, "sclk_ceiling_MHz": 2400, "idle_package_W": idle_w, "builds": rows_e,
           "reading": "with everything on the package sits at its limit and the firmware takes the shader clock down (2.0-2.15 GHz of 2.4); "
                      "compute only and memory only each stay below the limit at the full clock.  Energy per launch of the full kernel ~= "
                      "energy(memory only) + energy(compute only) - idle power x the time saved by overlapping: the kernel is bound by the "
                      "package power, not by HBM bandwidth, LDS or issue -- a faster build has to spend fewer joules per sample (DESIGN.md 4.1)"}
    json.dump(doc, open("profiles/%s_corr_energy.json" % tag, "w"), indent=1)
    for r in rows_e:
        print("energy:", r)
except (OSError, KeyError, ValueError, TypeError) as e:
    print("correlator energy summary skipped:", e)

# ---- round 6: the front-end claim sweep
try:
    sw = [json.loads(ln) for ln in open(os.path.join(src, "claim_sweep.jsonl")) if ln.startswith("{")]
    doc = {"what": "ms per pipelined step of the stock chain against the LDS claim of the front-end kernel (aisx_agc_set_lds_claim), by "
                   "channel count: the chain's own choice (aisx_chain.hip: chain_front_claim) and every claim of the sweep set by hand on "
                   "the same chain object, medians of three interleaved runs of 20 steps",
           "command": "python tools/claim_sweep.py --claims 0,16,24,32,40,48,56,63,72 --steps 20 --reps 3", "rows": sw}
    json.dump(doc, open("profiles/%s_claim_sweep.json" % tag, "w"), indent=1)
    for r in sw:
        print("claim sweep:", r["nchan"], r["chosen_claim_bytes"], r["chosen_over_best"])
except (OSError, KeyError, ValueError, TypeError) as e:
    print("claim sweep summary skipped:", e)
