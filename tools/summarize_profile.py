#!/usr/bin/env python3
"""Turn the rocprofv3 output of a `bench.py` run (gpurun_out/<dir>) into the committed
summaries under profiles/: per-kernel averages over the TIMED launches of the run (the
kernel trace is cut to the last K launches of the chain, so that warm-up and the isolated
correlator launches do not blur the average the bench line must agree with), the PMC
per-launch averages of one kernel, and bench.py's own JSON line."""
import csv
import glob
import json
import os
import sys


def kernel_rows(trace_csv):
    rows = []
    for r in csv.DictReader(open(trace_csv)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    return rows


def short(name):
    return name.replace("void ", "").split("(")[0]


def timed_stats(trace_csv, steps, warmup):
    """The default bench run = (warmup + steps) steps of the stock chain, 7 isolated correlator
    launches, then the same for the corr_est -> msk chain.  Stock-chain kernels are identified by
    position: launches warmup .. warmup+steps of each stock-only kernel; the correlator's timed
    launches are those that run while a k_fs_est of a timed step precedes them."""
    rows = kernel_rows(trace_csv)
    per = {}
    for s, e, n in rows:
        per.setdefault(short(n), []).append((s, e))
    out = {}
    nstock = warmup + steps
    for k, v in per.items():
        if not k.startswith("k_"):
            continue
        if len(v) < nstock:
            continue  # (a kernel of the side measurements only, e.g. the other correlator build)
        first = v[:nstock]  # the stock chain comes first in the run
        timed = first[warmup:nstock]
        if timed:
            d = [(e - s) / 1e6 for s, e in timed]
            out[k] = dict(n=len(d), avg=sum(d) / len(d), mn=min(d), mx=max(d), calls=len(v))
    return out


def pmc_avg(dirs, kernel_sub):
    acc = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if kernel_sub in r["Kernel_Name"]:
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


if __name__ == "__main__":
    src = sys.argv[1]
    tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
    line = None
    for ln in open(os.path.join(src, "bench_default.log")):
        if ln.startswith("{"):
            line = json.loads(ln)
    trace = glob.glob(os.path.join(src, "stats", "*", "*kernel_trace.csv"))[0]
    stats = glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))[0]
    st = timed_stats(trace, line["steps"], line["warmup"])
    os.makedirs("profiles", exist_ok=True)
    with open("profiles/%s_default_bench_kernel_stats.csv" % tag, "w") as f:
        f.write(open(stats).read())
    json.dump(line, open("profiles/%s_default_bench_line.json" % tag, "w"), indent=1)
    md = ["# rocprofv3 --kernel-trace --stats -- python bench.py   (MI355X, default run)", "",
          "Command on the GPU box: `cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats "
          "--output-format csv -d gpurun_out/<dir>/stats -- python bench.py`; summarised by `tools/summarize_profile.py`.", "",
          "bench.py's line of the same run (`%s_default_bench_line.json`): value %.0f complex MS/s, %.2f ms/step, "
          "`roofline.kernel_ms` %.3f (hipEvents over the timed region), `kernel_ms_alone` %.3f."
          % (tag, line["value"], line["ms_per_step"], line["roofline"]["kernel_ms"], line["roofline"]["kernel_ms_alone"]), "",
          "Per kernel, the %d TIMED launches of the whole-flowgraph chain only (the run also holds %d warm-up steps, "
          "7 isolated correlator launches and the corr_est -> msk-only chain; rocprofv3's own `--stats` table over "
          "ALL launches is in `%s_default_bench_kernel_stats.csv`):" % (line["steps"], line["warmup"], tag), "",
          "| kernel | timed launches | avg ms | min ms | max ms | launches in the whole run |", "|---|---|---|---|---|---|"]
    for k, v in sorted(st.items(), key=lambda kv: -kv[1]["avg"]):
        md.append("| %s | %d | %.3f | %.3f | %.3f | %d |" % (k[:48], v["n"], v["avg"], v["mn"], v["mx"], v["calls"]))
    kname = line["roofline"]["kernel"]
    for k in st:
        if k == kname or k.startswith(kname + "<"):
            md += ["", "`%s`: %.3f ms here against `roofline.kernel_ms` = %.3f ms in the bench line." % (
                k, st[k]["avg"], line["roofline"]["kernel_ms"])]
    open("profiles/%s_default_bench_kernel_stats.md" % tag, "w").write("\n".join(md) + "\n")
    print("\n".join(md))
