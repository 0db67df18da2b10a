#!/usr/bin/env python3
"""bench.py -- complex MS/s through the demod hot path on MI355X, with the
corr_est_cc kernel's HBM-roofline fraction, parity gates against the CPU oracle
and a CPU baseline.

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the chain over one batch: CHANNELS_PER_GPU channels x
SAMPLES complex samples per GPU, already resident in HBM, through the PRODUCT's
pipelined chain (ais_amd.ais_demod.work_pipelined = aisx_chain_step, include/aisx.h:
the path the config 3 / 4 / 5 tests run).  Weak scaling: every rank owns its own
channels, no data-path collective (channels are independent);
torch.distributed (RCCL) is used only for the timing barrier and the max / min
over ranks.  With --gpus N > 1 and no launcher environment the script starts its
N ranks itself (torch.distributed.run on 127.0.0.1), one process per GPU.

Workload (BASELINE.json configs[2], the one the metric is quoted on): 4096
batched channels, 65536 samples each, sps = 4, stock template (N = 896, SURVEY
D4).  The default chain is the whole flowgraph of python/ais_demod.py:56:
freq_sync (square -> FFT -> freqest -> NCO mix) -> feedforward agc -> corr_est
-> msk_timing_recovery -> NRZI bit tail; `value` is its throughput.  The same
run also times `--chain core` (corr_est -> msk_timing only, the two blocks the
metric string names) and the correlator alone at BASELINE config 2's shapes
(256 and 4096 channels, N = 896 and 112) and reports them under
"corr_est_to_msk_only" and "corr_only".  BASELINE config 4 (65536 channels on
8 GPUs) is `--gpus 8 --config4`.

"roofline" is the correlator's main kernel: algorithmic 16 B per sample over its
launch time by hipEvents on its own stream inside the timed region, next to the
8 TB/s spec peak and to what the best plain copy sustains on this box
("copy_ceiling_GBs").  After the timed region the last step of PARITY_CHANNELS
channels is compared with the CPU oracle replaying the same steps on the same
samples ("parity": tag offsets, peak magnitudes, time_est, decoded bursts --
BASELINE.md section 3).  "cpu_baseline": the oracle built on this host with -O3
-march=native, one thread and all hardware threads.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "gr-ais_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

# The pipelined step keeps four streams busy besides the default one (stream stages, timing
# recovery, bit tail, frequency estimates one step ahead); with the runtime's default of four
# hardware queues two of them would share a queue and run one after the other.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CORR_BYTES_PER_SAMPLE = 16  # 8 B read + 8 B delayed pass-through write
NUNIQ = 32  # CPU-generated channels the device input is built from
METRIC = "complex MS/s through corr_est->msk_timing chain; corr_est %HBM roofline"
CHAIN_TEXT = {"core": "corr_est->msk_timing+NRZI tail (no freq_sync / agc in front)",
              "stock": "freq_sync(freqest)->agc->corr_est->msk_timing+NRZI tail (python/ais_demod.py:56)",
              "corr": "corr_est only"}


def make_template(family, sps):
    import synth
    from ais_amd import gmsk_mod, modulate_vector_bc

    if family == "S":
        return modulate_vector_bc(gmsk_mod(sps, 0.4), [1, 1, 0, 0] * 7, [1])
    lv = [1 if b else -1 for b in synth.sync_bits("P")]
    return synth.gmsk_waveform(np.array(lv, float), sps)[: len(lv) * sps].astype(np.complex64)


def input_params(family, stock):
    return dict(amp=0.3 if stock else 1.0, cfo_max=500.0 if stock else (15.0 if family == "P" else 3.0))


def make_input(nchan, T, family, sps, device, rank, stock):
    """Synthetic IQ resident on the device: NUNIQ CPU-generated channels
    (seeded) replicated with a per-channel carrier phase, plus
    per-sample device-generated noise so that no two channels are equal."""
    import torch
    import synth

    ip = input_params(family, stock)
    amp = ip["amp"]
    base = np.stack([synth.make_channel(synth.SEED0 + 1000 * rank + c, T, family, sps, amp=amp, cfo_max=ip["cfo_max"],
                                        noise=False)[0] for c in range(NUNIQ)])
    g = torch.Generator(device=device)
    g.manual_seed(synth.SEED0 + rank)
    b = torch.as_tensor(base).to(device)
    reps = (nchan + NUNIQ - 1) // NUNIQ
    x = b.repeat(reps, 1)[:nchan].contiguous()
    ph = torch.rand(nchan, generator=g, device=device) * (2 * np.pi)
    x *= torch.polar(torch.ones_like(ph), ph).to(torch.complex64).view(-1, 1)
    sigma = amp * np.sqrt(sps / (10 ** (20.0 / 10.0)) / 2.0)  # Eb/N0 = 20 dB
    noise = torch.randn((nchan, T, 2), generator=g, device=device, dtype=torch.float32) * sigma
    x += torch.view_as_complex(noise)
    return x


def burst_infos(c, T, family, sps, rank, stock):
    """What was transmitted in device channel c (= base channel c % NUNIQ)."""
    import synth

    ip = input_params(family, stock)
    return synth.make_channel(synth.SEED0 + 1000 * rank + (c % NUNIQ), T, family, sps, amp=ip["amp"], cfo_max=ip["cfo_max"],
                              noise=False)[1]


def pmc_traffic(nchan, T, N):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC
    summary (collected in separate --pmc passes, see profiles/), if it was taken on
    this workload; (None, None) otherwise."""
    import glob

    kern = corr_kernel_name(N)
    # (the newest round's summary of THIS kernel on THIS workload: rNN_corr*_pmc.json, highest NN first)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_corr*_pmc.json")), reverse=True):
        name = os.path.basename(path)
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        w = d.get("workload", {})
        if not str(d.get("kernel", "")).startswith(kern):
            continue
        if (w.get("channels"), w.get("samples"), w.get("template_len")) == (nchan, T, N):
            return d.get("hbm_bytes_per_launch"), "profiles/%s (kernel %s; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % (
                name, d.get("kernel"))
    return None, None


def energy_split():
    """The committed energy split of the F = 4096 correlator alone (profiles/rNN_corr_energy.json, tools/corr_energy.py:
    time x package power of the product build, a compute-only and a memory-only build), newest round first; None if absent."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_corr_energy.json")), reverse=True):
        try:
            d = json.load(open(path))
            b = {r["build"]: r for r in d["builds"]}
            f, c, m = b["f"], b["fdbg11"], b["fdbg88"]
            lim = float(d.get("package_limit_W", 1400))
            return {"source": "profiles/%s (static: measured by tools/corr_energy.py, not in this run)" % os.path.basename(path),
                    "package_limit_W": lim,
                    "joule_per_launch": {"product": f["joule_per_launch"], "compute_only": c["joule_per_launch"],
                                         "memory_only": m["joule_per_launch"]},
                    "ms": {"product": f["kernel_ms"], "compute_only": c["kernel_ms"], "memory_only": m["kernel_ms"]},
                    "package_W": {"product": f["package_W"], "compute_only": c["package_W"], "memory_only": m["package_W"]},
                    "sclk_MHz": {"product": f["sclk_MHz"], "compute_only": c["sclk_MHz"], "memory_only": m["sclk_MHz"]},
                    "ms_at_the_limit": (c["joule_per_launch"] + m["joule_per_launch"]) / lim * 1e3,
                    "reading": "joules(product) ~= joules(compute only) + joules(memory only): at the package limit a launch can not "
                               "take less than that sum / limit (ms_at_the_limit) -- the kernel alone is bound by package power, not by "
                               "HBM bandwidth; frac 0.60 of 8 TB/s (0.895 ms) needs about a third fewer joules in the transform (DESIGN.md 4.1)"}
        except (OSError, ValueError, KeyError, TypeError):
            continue
    return None


MSK_BYTES_PER_SAMPLE = 10.25  # 8 read + 8 / sps symbol + 1 / sps bit written, sps = 4


def probe_gnuradio():
    """Is the reference's own CPU path (GNU Radio 3.8 + VOLK) present on THIS host?  Looks for the
    things its build needs (CMakeLists.txt:71 find_package(Gnuradio "3.8")): the Python package, the config tool,
    VOLK's profiler / pkg-config entry, the CMake package files.  Says what it found; never builds anything."""
    import importlib.util
    import shutil

    found = []
    try:
        # (gnuradio.gr, not gnuradio: this repository's own gr-ais_amd/gnuradio/ directory -- the wrapper sources --
        # is on sys.path and would pass for a namespace package of that name)
        if importlib.util.find_spec("gnuradio.gr") is not None:
            found.append("python package gnuradio.gr")
    except (ImportError, ValueError, AttributeError):
        pass
    for tool in ("gnuradio-config-info", "volk_profile", "volk-config-info"):
        w = shutil.which(tool)
        if w:
            found.append(w)
    for d in ("/usr/lib/x86_64-linux-gnu/cmake/gnuradio", "/usr/lib/cmake/gnuradio", "/usr/local/lib/cmake/gnuradio",
              "/usr/lib/x86_64-linux-gnu/cmake/volk", "/usr/local/lib/cmake/volk", "/usr/include/volk", "/usr/local/include/volk",
              "/usr/include/gnuradio", "/usr/local/include/gnuradio"):
        if os.path.isdir(d):
            found.append(d)
    try:
        pc = subprocess.run(["pkg-config", "--modversion", "gnuradio-runtime", "volk"], capture_output=True, text=True, timeout=10)
        if pc.returncode == 0:
            found.append("pkg-config: " + pc.stdout.strip().replace("\n", " / "))
    except (OSError, subprocess.SubprocessError):
        pass
    if not found:
        return ("unavailable: probed this host for the gnuradio Python package, gnuradio-config-info, volk_profile, "
                "volk-config-info, the gnuradio / volk CMake and include directories and pkg-config entries -- none present, "
                "so /root/reference (find_package(Gnuradio \"3.8\")) can not be built or timed here")
    return "found on this host, not timed (the reference still needs its own build): " + "; ".join(found)


def measure_h2d(torch, device):
    """Host-to-device rate of a pinned 1 GiB buffer (hipMemcpyAsync through torch, best of three): a caller that hands
    over HOST buffers of IQ is bound by it -- 8 bytes per complex sample in; the decoded bits coming back are 1 / 32 of that."""
    try:
        n = 1 << 30
        h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        d = torch.empty(n, dtype=torch.uint8, device=device)
        best = 0.0
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d.copy_(h, non_blocking=True)
            torch.cuda.synchronize()
            best = max(best, n / (time.perf_counter() - t0) / 1e9)
        del h, d
        return {"h2d_GBs": best, "h2d_bound_MSs": best * 1e9 / 8.0 / 1e6,
                "note": "pinned host memory -> HBM, 1 GiB; the headline value has its input resident in HBM -- a host-buffer "
                        "hand-over is bound by h2d_bound_MSs per GPU"}
    except RuntimeError as e:
        return {"h2d_GBs": None, "h2d_bound_MSs": None, "note": "not measured: %s" % e}


def hbm_ceilings():
    """tools/ubench/hbm_ceiling (built by __graft_entry__.build()): read-only, write-only and copy rates by access shape."""
    exe = os.path.join(ROOT, "tools", "ubench", "hbm_ceiling")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "json"], capture_output=True, text=True, timeout=120)
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except (OSError, ValueError, IndexError, subprocess.SubprocessError):
        return None
    return {"best_copy_GBs": d.get("best_copy_GBs"), "best_read_GBs": d.get("best_read_GBs"), "best_write_GBs": d.get("best_write_GBs"),
            "copy_128MiB_GBs": next((r["GBs"] for r in d.get("rows", []) if r["GiB"] < 0.2), None),
            "source": "tools/ubench/hbm_ceiling.hip: 2 GiB buffers (beyond the 256 MiB Infinity Cache), hipEvents over 10 launches, "
                      "best of the shapes swept; copy_128MiB_GBs is what a cache-resident size reads (the guide's 6.29 TB/s class)"}


def config1_host_path():
    """BASELINE config 1 (one 48 kS/s channel) through the gr::ais block classes' work() / general_work() with HOST buffers
    -- the GNU Radio drop-in path -- as a throughput: tests/abi_cpp/gr_blocks_harness.cpp (the compiled scheduler stand-in
    of tests/test_gr_wrappers.py) run over tests/golden/config1_sched.bin, warm (second of two runs)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import test_gr_wrappers as tw

        exe = tw._build()
        best = None
        for _ in range(2):
            out = subprocess.run([exe, tw.FIXTURE], capture_output=True, text=True, timeout=300)
            for ln in out.stdout.splitlines():
                if ln.startswith("HOST_PATH"):
                    kv = dict(w.split("=") for w in ln.split()[1:])
                    best = dict(samples=int(kv["samples"]), seconds=float(kv["seconds"]), MSs=float(kv["MSs"]),
                                passed="PASS" in out.stdout)
        if best is None:
            return None
        best["what"] = ("one channel, freq_sync -> agc -> corr_est -> msk through make() / work() / general_work() with host pointers "
                        "(every call: copy in, launch, copy out, synchronous); the stock application needs 2 x 0.05 MS/s "
                        "(python/radio.py:88-89,120); one CPU thread of the port runs cpu_baseline.single_thread_value")
        return best
    except Exception as e:  # noqa: BLE001  (a side measurement must not take the line with it)
        return {"MSs": None, "note": "not measured: %s" % e}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(chain, family, sps, T, budget_s=6.0):
    """The CPU path timed beside the GPU's, on this host, on a bounded sample of the same workload.
    The reference's own VOLK / FFTW path needs GNU Radio (absent), so this is the oracle: a plain-C
    port of the reference's algorithm (`kind: "port"`).  Timed build = oracle/ais_oracle.c compiled
    HERE with -O3 -march=native, work buffers kept between channels (a
    flowgraph allocates its buffers once), -ffp-contract=off kept so that its results are the
    portable build's, which is checked first (orc_demod_hash).
      B1  one thread, whole channels of T samples one after the other (the like-for-like of one
          GNU Radio flowgraph of one channel);
      B2  one worker thread per hardware thread, each running whole channels.
    `value` is the better of B2 and the same with one thread per two hardware threads, `cores` the threads
    that run used; B1 and the portable -O2 build's figures (what rounds 1-2 reported) stand next to it."""
    import tempfile

    import oracle_py as orc
    import synth

    tmpl = make_template(family, sps)
    stock = chain == "stock"
    ip = dict(amp=0.3 if stock else 1.0, cfo_max=500.0 if stock else 15.0)
    stages = 3 if stock else 0
    xs = np.stack([synth.make_channel(synth.SEED0 + c, T, family, sps, **ip)[0] for c in range(8)])
    ncores = os.cpu_count() or 1
    try:
        ncores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    tmpdir = tempfile.mkdtemp(prefix="aisx_orc_")
    native = orc.build_native(os.path.join(tmpdir, "libais_oracle_native.so"))
    same = None
    if native is not None:
        same = all(orc.demod_hash(sps, tmpl, stages, xs[c]) == orc.demod_hash(sps, tmpl, stages, xs[c], native) for c in (0, 1))
        if not same:
            native = None  # (a timing build whose results differ is not the same algorithm: do not time it)

    def run(L, nthreads, budget):
        done, wall = orc.demod_bench_mt(nthreads, sps, tmpl, stages, xs, budget, L)
        return done * T / wall / 1e6, done, wall

    b1, n1, w1 = run(native, 1, budget_s)
    b2, n2, w2 = run(native, ncores, budget_s)
    half = run(native, max(1, ncores // 2), budget_s / 2)[0]  # (one thread per core if SMT is on)
    p1 = run(None, 1, budget_s / 2)[0]
    p2 = run(None, ncores, budget_s / 2)[0]
    flags = "gcc -O3 -march=native -ffp-contract=off, buffers kept between channels" if native is not None else \
        "gcc -O2 (no native build on this host%s)" % ("" if same is None else ": its results differed")
    note = None
    if b2 / b1 < 30.0 and ncores >= 64:
        note = ("B2 / B1 = %.1f on %d threads (%.1f on %d): every channel streams ~%.0f MB through buffers far larger than a "
                "core's share of L2/L3 (input 0.5 MB + five work buffers of 0.5 MB + 6 MB of tag scratch per 65536 samples), "
                "the per-sample work is short dependent float chains (NCO phase, timing loop) that SMT siblings share one "
                "core's ports for, and the radix-2 FFT walks its 16 KB block twelve times" %
                (b2 / b1, ncores, half / b1, max(1, ncores // 2), 8.0 * T * 7 / 1e6))
    best, best_cores = (b2, ncores) if b2 >= half else (half, max(1, ncores // 2))
    return dict(value=best, unit="complex MS/s", cores=best_cores, kind="port", all_threads_value=b2,
                sample="B2: %d channels x %d samples on %d threads in %.1f s wall; B1: %d channels on one thread (%.1f s); chain=%s; "
                       "oracle/ais_oracle.c (%s; radix-2 FFT standing in for FFTW/VOLK)" % (n2, T, ncores, w2, n1, w1, chain, flags),
                single_thread_value=b1, scaling_B2_over_B1=b2 / b1, half_threads_value=half,
                native_build=native is not None, native_results_equal_portable=same,
                portable_O2_build={"single_thread_value": p1, "value": p2},
                scaling_note=note, cpu_model=cpu_model(), nproc=os.cpu_count(),
                reference_volk_path=probe_gnuradio())


def oracle_replay(chain, tmpl, sps, xk, nsteps, max_noutput=0):
    """The oracle stepping `nsteps` times over the same samples (one row of xk per channel), as the
    benchmark does; returns, per channel, the last step's (bits, tags).  One thread per channel."""
    import concurrent.futures as cf

    import oracle_py as orc

    orc.lib()

    def run(c):
        if chain == "corr":
            o = orc.CorrEst(tmpl, float(sps), 1, 0.9)
            for _ in range(nsteps):
                _, _, tags = o.work(xk[c])
            return None, tags
        dem = orc.Demod(sps, tmpl, stages=3 if chain == "stock" else 0, max_noutput=max_noutput)
        for _ in range(nsteps):
            bits, _, tags = dem.step(xk[c])
        return bits, tags

    with cf.ThreadPoolExecutor(min(len(xk), os.cpu_count() or 1)) as ex:
        return list(ex.map(run, range(len(xk))))


def parity_gates(chain, tmpl, sps, T, family, rank, xk, nsteps, gpu_tags, gpu_bits, gpu_prod, thresh, max_noutput=0):
    """the last step of the channels in xk, HIP path vs oracle."""
    from parity import compare_bursts, compare_detections

    t0 = time.perf_counter()
    ref = oracle_replay(chain, tmpl, sps, xk, nsteps, max_noutput)
    K = len(xk)
    tot = dict(detections=0, matched=0, offsets_equal=0, lone=0, lone_near_threshold=0)
    mag = tim = 0.0
    ncmp = same = near = 0
    count_equal = 0
    for c in range(K):
        obits, otags = ref[c]
        r = compare_detections(gpu_tags[gpu_tags["chan"] == c], otags, thresh)
        for k in tot:
            tot[k] += r[k]
        mag = max(mag, r["mag_rel_max"])
        tim = max(tim, r["time_est_abs_max"])
        if obits is not None:
            gb = gpu_bits[c, : gpu_prod[c]]
            count_equal += int(gpu_prod[c] == len(obits))
            a, b, d = compare_bursts(gb, obits, burst_infos(c, T, family, sps, rank, chain == "stock"))
            ncmp, same, near = ncmp + a, same + b, near + d
    out = {"channels": K, "steps_replayed": nsteps,
           "detections": tot["detections"], "detections_matched_within_1": tot["matched"], "offsets_equal": tot["offsets_equal"],
           "tags_within_1": bool(tot["lone"] == tot["lone_near_threshold"]),
           "seen_by_one_side_only": tot["lone"], "of_which_within_2e-5_of_threshold": tot["lone_near_threshold"],
           "mag_rtol_max": mag, "time_est_abs_max": tim,
           "mag_gate_1e-5": bool(mag <= 1e-5), "time_est_gate_1e-4": bool(tim <= 1e-4),
           "oracle_seconds": round(time.perf_counter() - t0, 2)}
    if chain != "corr":
        # bursts_within_4: the burst's bits are in the HIP stream within +-4 bit positions of where the oracle has
        # them (a time_est that differs in its last place can slip one symbol in the noise before a burst);
        # bursts_in_place: at exactly the same position
        out.update(bursts_compared=ncmp, bursts_identical=near, bursts_identical_same_position=same,
                   bursts_in_place=same, bursts_within_4=near, symbol_counts_equal=count_equal)
    return out


def bench_wideband(args, torch, device, as_dict=False, steps=None):
    """BASELINE config 5 (one GPU): 25 MS/s wideband IQ -> 1024-lane polyphase channelizer
    (2x oversampled: 48.83 kS/s per lane = 5.086 samples/symbol) -> corr_est -> msk chain
    on the 1024 lanes with the sps = 5 template (the stock app runs 5.2083 sps against a
    5 sps template, python/radio.py:49-57).  No reference number or parity target exists
    for the channelizer beyond the per-channel filter it replaces (tests/test_pfb.py)."""
    import ais_amd

    fs, M, D = 25e6, 1024, 512
    nfr = 32768
    n = nfr * D
    g = torch.Generator(device=device)
    g.manual_seed(7)
    x = torch.view_as_complex(torch.randn((1, n, 2), generator=g, device=device, dtype=torch.float32) * 0.1)
    taps = ais_amd.firdes_low_pass(1.0, fs, 11e3, 1e3)
    pfb = ais_amd.pfb_channelizer_ccf(M, taps, decim=D, max_frames=nfr)
    sps = fs / D / 9600.0
    tmpl = make_template("S", 5)
    opts = dict(samples_per_symbol=sps, bits_per_sec=9600.0, clockrec_gain=0.04, omega_relative_limit=0.01, fftlen=1024)
    dem = ais_amd.ais_demod(opts, nchan=M, max_items=nfr, stages="core", preamble_symbols=tmpl[:1024])

    def step():
        lanes = pfb.work(x)
        dem.work(lanes)

    nsteps = steps or args.steps
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t0 = time.perf_counter()
    for _ in range(nsteps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ev[0].record(); lanes = pfb.work(x); ev[1].record(); dem.work(lanes); ev[2].record()
    torch.cuda.synchronize()
    line = {
        "metric": "wideband complex MS/s through polyphase channelizer -> 1024 demod lanes",
        "value": n * nsteps / el / 1e6, "unit": "complex MS/s (wideband input)", "n_gpus": 1, "steps": nsteps,
        "warmup": args.warmup, "ms_per_step": el / nsteps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "1 x %d wideband samples/step at 25 MS/s -> 1024 lanes x %d items (decim 512, 60227-tap "
                               "prototype) -> corr_est(N=1024)->msk" % (n, nfr)},
        "pfb_ms": ev[0].elapsed_time(ev[1]), "demod_ms": ev[1].elapsed_time(ev[2]),
        "realtime_factor": n * nsteps / el / 25e6}
    del pfb, dem, x
    torch.cuda.empty_cache()
    if as_dict:
        return line
    print(json.dumps(line))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--channels-per-gpu", type=int, default=4096,
                    help="channels each rank owns (weak scaling); BASELINE config 4 = --gpus 8 --channels-per-gpu 8192")
    ap.add_argument("--config4", action="store_true",
                    help="BASELINE config 4's per-GPU shape: 8192 channels per GPU (65536 channels on 8 GPUs); "
                         "the same as --channels-per-gpu 8192, named in config.workload")
    ap.add_argument("--samples", type=int, default=65536)
    ap.add_argument("--template", choices=["S", "P"], default="S", help="S: stock 896-sample template; P: 112-sample preamble")
    ap.add_argument("--chain", choices=["core", "stock", "corr", "wideband"], default="stock",
                    help="stock = the whole ais_demod.py flowgraph (freq_sync with freqest, agc, corr_est, msk timing "
                         "recovery, NRZI tail); core = corr_est -> msk only; corr = corr_est only; wideband = BASELINE "
                         "config 5: one 25 MS/s stream -> 1024-lane polyphase channelizer -> core chain")
    ap.add_argument("--single-chain", action="store_true",
                    help="only the chain asked for: no corr_est->msk-only and correlator-only side measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-channels", type=int, default=8, help="channels of the last step compared with the oracle (0: off)")
    ap.add_argument("--dry", action="store_true",
                    help="launcher check without a GPU: the ranks rendezvous over gloo, shard the channels and rank 0 "
                         "prints the line with value 0 (tests/test_multiproc.py)")
    return ap.parse_args(argv)


def spawn_ranks(args):
    """--gpus N without a launcher: start N ranks of this script, one per GPU, on this node."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def corr_kernel_name(n):
    """the correlator kernel aisx_corr_process launches for an n-item template (aisx_lib.hip)"""
    return "k_corr2d_main" if n <= 512 else "k_corr4f_main"


class PowerSampler:
    """Package power and shader clock of one GPU from its hwmon files (power1_average / power1_input in uW, freq1_input in
    Hz), read every few ms by a thread while a measurement runs.  Why it is in the line: the correlator alone runs the
    package into its 1400 W limit and the firmware answers with the shader clock (2.0-2.15 GHz instead of 2.4), and the
    kernel's time follows the clock -- the number next to a roofline fraction that says which limit was met."""

    def __init__(self, index=0, period=0.004):
        import glob

        self.files = None
        cards = []
        for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            pw = next((os.path.join(h, f) for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(h, f))), None)
            fq = os.path.join(h, "freq1_input")
            if pw:
                pci = os.path.basename(os.path.realpath(os.path.join(h, "..", "..")))  # 0000:bb:dd.f
                cards.append((pw, fq if os.path.exists(fq) else None, pci))
        # the card of THIS process's device: by PCI address (a box may show more cards in sysfs than the process may use)
        want = None
        try:
            import torch

            pr = torch.cuda.get_device_properties(index)
            want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception:  # noqa: BLE001
            pass
        pick = [c for c in cards if want and c[2].lower().startswith(want)]
        if pick:
            self.files = pick[0][:2]
        elif len(cards) == 1:
            self.files = cards[0][:2]
        self.period = period
        self.samples = []
        self._stop = None
        self._th = None

    def _read(self, path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def start(self):
        import threading

        if self.files is None:
            return self
        self.samples = []
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                pw = self._read(self.files[0])
                fq = self._read(self.files[1]) if self.files[1] else None
                self.samples.append((pw, fq))
                self._stop.wait(self.period)

        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()
        return self

    def stop(self):
        if self._th is None:
            return None
        self._stop.set()
        self._th.join()
        self._th = None
        pw = [a / 1e6 for a, _ in self.samples if a]
        fq = [b / 1e6 for _, b in self.samples if b]
        if not pw:
            return None
        return {"package_W_mean": float(np.mean(pw)), "package_W_max": float(np.max(pw)),
                "sclk_MHz_mean": float(np.mean(fq)) if fq else None, "sclk_MHz_min": float(np.min(fq)) if fq else None,
                "samples": len(pw), "source": self.files[0]}


def spin_up(torch, device, ms=250.0):
    """Untimed device work before the warm-up steps: after idle the firmware ramps the shader clock over tens of ms
    (0.6 -> 2.1 GHz in ~60 ms on this pool), and every kernel of this path is clock-bound -- a 20-step timed region
    behind five warm-up steps would otherwise start on a chip that is still coming up.  A streaming receiver never idles."""
    a = torch.empty(64 << 20, dtype=torch.float32, device=device)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(20):
            a.mul_(1.0001)
        torch.cuda.synchronize()
    del a


def main():
    args = parse_args()
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ  # under torch.distributed.run
    if args.gpus > 1 and not launched:
        raise SystemExit(spawn_ranks(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    if launched and world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d ranks; reporting n_gpus = %d" % (args.gpus, world, world),
              file=sys.stderr)
    if args.config4:
        args.channels_per_gpu = 8192
    sps, T, nchan = 4, args.samples, args.channels_per_gpu
    from ais_amd.shard import max_over_ranks, shard_channels

    if args.dry:
        if launched:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo")
        first, cnt = shard_channels(nchan * world, world, rank)
        assert cnt == nchan and first == rank * nchan
        if launched:
            dist.barrier()
        el = max_over_ranks(1e-3 * (1 + rank))
        el_min = -max_over_ranks(-1e-3 * (1 + rank))
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "complex MS/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "f32", "data": "none", "dry": True,
                              "config": {"workload": "dry run: %d ranks x %d channels%s, nothing computed" % (
                                  world, nchan, " (BASELINE config 4: %d channels in all)" % (world * nchan) if args.config4 else ""),
                                         "channels_per_gpu": nchan, "parallelism": "channel-sharded x%d, no collective" % world},
                              "max_over_ranks_check": el,
                              "rank_ms_per_step": {"min": 1e3 * el_min, "max": 1e3 * el}}))
        if launched:
            dist.barrier()
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libaisx has no CPU path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = launched
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)  # "nccl" is RCCL on ROCm

    import ais_amd

    if args.chain == "wideband":
        return bench_wideband(args, torch, device)
    tmpl = make_template(args.template, sps)
    opts = dict(samples_per_symbol=sps, bits_per_sec=9600.0, clockrec_gain=0.04, omega_relative_limit=0.01, fftlen=1024)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(chain, want_parity, lookahead=True, msk_tp=False, msk_Q=0, nch=None, corr_claim=0):
        """K timed steps of `chain` on this rank's channel shard; returns the wall time (max over
        ranks), the correlator kernel's per-launch times inside the timed region and (rank 0) the
        parity gates of the last step.  The step is the product's pipelined chain
        (ais_demod.work_pipelined = aisx_chain_step, include/aisx.h): the sample passes of step k + 1
        on one stream beside the timing recovery of step k on another, its bit tail and the NCO
        phase walk of step k + 2 on two more.  This benchmark feeds the same samples every step, so
        the next step's input is always at hand (x_next = x); a live source runs one block ahead."""
        stock = chain == "stock"
        nchan = nch or args.channels_per_gpu
        x = make_input(nchan, T, args.template, sps, device, rank, stock)
        # (the buffers of the alone-on-the-chip measurement at the end are allocated now: a 2 GB buffer
        # allocated late comes out of what the allocator has left over, and the same kernel reads
        # ~20 % longer on it -- placement, not the kernel)
        early = os.environ.get("AISX_BENCH_EARLY_ISO") == "1"
        yout = torch.empty((nchan, T), dtype=torch.complex64, device=device) if early else None
        yin_buf = torch.empty((nchan, T + 1024), dtype=torch.complex64, device=device) if (stock and early) else None
        dem = ais_amd.ais_demod(opts, nchan=nchan, max_items=T, stages="stock" if stock else "core",
                                preamble_symbols=tmpl, fused_front_end=True)
        corr = dem.preamble_detect
        corr.set_profiling(True)
        if corr_claim:
            corr.set_lds_claim(corr_claim)
        if chain != "corr":
            if msk_Q:
                dem.clockrec.set_max_noutput_items(msk_Q)
            if msk_tp:
                dem.clockrec.set_time_parallel(64, 1, 16384)
        y_corr = [torch.empty((nchan, T), dtype=torch.complex64, device=device) for _ in range(2)] if chain == "corr" else None
        state = dict(k=0, last=None)

        def step():
            if chain == "corr":
                corr.work(x, out=y_corr[state["k"] & 1])
            else:
                state["last"] = dem.work_pipelined(x, x_next=x if (stock and lookahead) else None)
            state["k"] += 1

        spin_up(torch, device)  # (untimed, not a step: the clocks of an idle chip take tens of ms to come up)
        for _ in range(args.warmup):
            step()
        barrier()
        corr.set_profiling(True)  # restart the event ring: the timed steps only
        if chain != "corr" and not msk_tp:
            dem.clockrec.set_profiling(True)
        ps = PowerSampler(local_rank).start()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        power = ps.stop()
        barrier()
        # per-launch duration of the dominant kernel over the timed region: hipEvents
        # recorded around it on its launch stream in every step, read back only now
        kern_ms = corr.kernel_ms_history()[-args.steps:]
        res = dict(kern_ms=kern_ms, st=0, ndet=0, parity=None, tag_overflow=False, nchan=nchan, power=power)
        res["msk_ms"] = dem.clockrec.kernel_ms_history()[-args.steps:] if (chain != "corr" and not msk_tp) else None
        if rank == 0:
            # the last step's results, before anything else touches the handles
            res["st"] = dem.clockrec.last_status() if chain != "corr" else 0
            try:
                tags = corr.tags()
            except OverflowError:
                res["tag_overflow"] = True
                tags = corr.tags(allow_overflow=True)
            res["ndet"] = int((tags["key"] == 2).sum())
            K = min(args.parity_channels, nchan) if want_parity else 0
            if K > 0:
                last = state["last"]
                gbits = last["bits"][:K].cpu().numpy() if chain != "corr" else None
                gprod = last["produced"][:K].cpu().numpy() if chain != "corr" else None
                res["parity"] = parity_gates(chain, tmpl, sps, T, args.template, rank, x[:K].cpu().numpy(),
                                             args.warmup + args.steps, tags[tags["chan"] < K], gbits, gprod, corr.threshold(), msk_Q)
            if msk_tp:
                res["restart_stats"] = dem.clockrec.restart_stats()
        barrier()
        # the same kernel with the chip to itself (no timing-recovery kernel alongside), on the
        # data it sees in the chain: behind the front end for the stock chain
        yin = x
        if stock:
            fs2 = ais_amd.square_and_fft_sync_cc(sps * 9600.0, 9600.0, 1024, nchan=nchan, max_items=T)
            agc2 = ais_amd.feedforward_agc_cc(512, 2.0, nchan=nchan, max_items=T + 1024)
            yin = ais_amd.freq_sync_agc(fs2, agc2, x, out=yin_buf)[0]
            del fs2, agc2
        if yout is None:
            yout = torch.empty((nchan, T), dtype=torch.complex64, device=device)
        # (launches back to back for 0.15 s, then twenty timed ones: a launch that follows a host synchronisation
        # starts on a clocked-down chip -- the shader clock takes tens of ms to come up -- and the kernel's time follows the clock)
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < 0.15:
            for _ in range(10):
                corr.work(yin, out=yout)
            torch.cuda.synchronize()
        for _ in range(5):
            corr.work(yin, out=yout)
        corr.set_profiling(True)
        ps_iso = PowerSampler(local_rank).start()
        for _ in range(20):
            corr.work(yin, out=yout)
        torch.cuda.synchronize()
        res["power_alone"] = ps_iso.stop()
        iso = corr.kernel_ms_history()[-20:]
        res["el"] = max_over_ranks(el, device=device)
        res["el_min"] = -max_over_ranks(-el, device=device)
        res["iso"] = iso
        del dem, x, y_corr, yin, yout, yin_buf
        torch.cuda.empty_cache()
        return res

    def measure_corr_only(nch, family):
        """corr_est alone (BASELINE config 2's shape when nch = 256): kernel time by hipEvents."""
        tm = make_template(family, sps)
        x = make_input(nch, T, family, sps, device, rank, False)
        blk = ais_amd.corr_est_cc(tm, float(sps), 1, 0.9, nchan=nch, max_items=T)
        blk.set_profiling(True)
        out = torch.empty_like(x)
        for _ in range(3):
            blk.work(x, out=out)
        torch.cuda.synchronize()
        # (a) as rounds 1-5 measured it: 20 launches behind a host synchronisation -- the shader clock is still coming up
        blk.set_profiling(True)
        for _ in range(20):
            blk.work(x, out=out)
        torch.cuda.synchronize()
        cold = float(np.mean(blk.kernel_ms_history()[-20:]))
        # (b) the steady state: launches back to back for 0.25 s (the clock has settled: spin_up), then 40 timed ones
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
            for _ in range(20):
                blk.work(x, out=out)
            torch.cuda.synchronize()
        for _ in range(10):
            blk.work(x, out=out)
        blk.set_profiling(True)
        ps = PowerSampler(local_rank).start()
        t0 = time.perf_counter()
        nrun = 40
        for _ in range(nrun):
            blk.work(x, out=out)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / nrun
        power = ps.stop()
        kms = float(np.mean(blk.kernel_ms_history()[-nrun:]))
        nbytes = CORR_BYTES_PER_SAMPLE * float(nch) * T
        r = dict(channels=nch, template_len=int(tm.size), kernel=corr_kernel_name(tm.size),
                 kernel_ms=kms, call_ms=wall * 1e3, achieved_GBs=nbytes / (kms * 1e-3) / 1e9,
                 frac=nbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, complex_MSs=float(nch) * T / wall / 1e6,
                 kernel_ms_first_20_after_sync=cold, frac_first_20_after_sync=nbytes / (cold * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 power=power,
                 how="steady state: 40 launches behind 0.25 s of back-to-back launches (settled clock); *_first_20_after_sync = the "
                     "20 launches behind a host synchronisation that rounds 1-5 reported (clock still ramping)")
        del blk, x, out
        torch.cuda.empty_cache()
        return r

    r = measure(args.chain, True)
    el, kern_ms, iso, st = r["el"], r["kern_ms"], r["iso"], r["st"]
    side = not args.single_chain
    # the default run also times the two-block chain the metric string names, and the correlator alone
    extra = measure("core", False) if (args.chain == "stock" and side) else None
    # the caveats of the headline as numbers: the same steps without the one-buffer look-ahead (x_next = None: every
    # step estimates for itself), and with the time-parallel timing recovery (opt-in, include/aisx.h)
    nola = measure("stock", False, lookahead=False) if (args.chain == "stock" and side and world == 1) else None
    # the placement choice the chain leaves to its caller: the correlator kept off the CUs that hold a recovery workgroup
    # (aisx_corr_set_lds_claim: a shorter step for a slower graded kernel, include/aisx.h)
    coff = measure("stock", False, corr_claim=17408) if (args.chain == "stock" and side and world == 1 and nchan <= 4096) else None
    tpm = measure(args.chain, True, msk_tp=True, msk_Q=256) if (args.chain != "corr" and side and world == 1) else None
    tpq = measure(args.chain, False, msk_tp=False, msk_Q=256) if (args.chain != "corr" and side and world == 1) else None
    corr_only = None
    if side and world == 1:
        corr_only = [measure_corr_only(c, f) for c in (256, 4096) for f in ("S", "P")]

    # BASELINE config 4's per-GPU shape (8192 channels) as a side measurement of the default run: there the
    # timing recovery hides behind the sample passes, and every millisecond taken off those shows one to one
    c4 = measure("stock", False, nch=8192) if (args.chain == "stock" and side and world == 1 and nchan != 8192) else None
    h2d = measure_h2d(torch, device) if (rank == 0 and side and world == 1) else None
    ceil = hbm_ceilings() if (rank == 0 and side and world == 1) else None
    copy_gbs = None
    if rank == 0 and side:
        # what a plain 16-byte-per-lane copy sustains on this chip, no profiler attached (2 GiB each way)
        from ais_amd import _lib as _L
        import ctypes as _C

        g = _C.c_float(0)
        if _L.lib().aisx_util_copy_GBs(2 << 30, 10, _C.byref(g)) == 0:
            copy_gbs = float(g.value)
    if rank == 0:
        total_samples = float(nchan) * T * world * args.steps
        kms = float(np.mean(kern_ms))
        achieved = CORR_BYTES_PER_SAMPLE * float(nchan) * T / (kms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(nchan, T, int(tmpl.size))
        line = {
            "metric": METRIC,
            "value": total_samples / el / 1e6,
            "unit": "complex MS/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3,
            "rank_ms_per_step": {"min": r["el_min"] / args.steps * 1e3, "max": el / args.steps * 1e3},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s%d batched channels/GPU x %d complex samples/step, sps=4, template N=%d (%s), chain=%s; "
                            "device-resident input (the same buffer every step), one-buffer look-ahead (the next step's "
                            "frequency estimates are prepared during this one: config.side.no_lookahead_ms_per_step is the same "
                            "work without), clocks spun up by 0.25 s of untimed device work before the warm-up steps"
                % ("BASELINE config 4 (%d channels on %d GPUs): " % (nchan * world, world) if args.config4 else "",
                   nchan, T, tmpl.size, "stock ais_demod.py" if args.template == "S" else "28-symbol preamble",
                   CHAIN_TEXT[args.chain]),
                "input": "device-resident", "lookahead_buffers": 1 if args.chain == "stock" else 0, "spin_up_ms": 250,
                "channels_per_gpu": nchan,
                "samples_per_step": T,
                "template_len": int(tmpl.size),
                "chain": args.chain,
                "parallelism": "channel-sharded x%d, no collective" % world,
            },
            "roofline": {
                "kernel": corr_kernel_name(tmpl.size),
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "copy_ceiling_GBs": copy_gbs,
                "frac_of_copy_ceiling": (achieved / copy_gbs) if copy_gbs else None,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_static": True,  # (read from the committed PMC summary of this workload, not collected in this run)
                "traffic_source": traffic_src,
                "kernel_ms": kms,
                "kernel_ms_alone": float(np.mean(iso)),
                "power": {"timed_region": r.get("power"), "kernel_alone": r.get("power_alone"),
                          "note": "package power (W) and shader clock (MHz) from the GPU's hwmon files while the measurement ran; "
                                  "the package limit is 1400 W, the clock's ceiling 2400 MHz: the correlator alone sits at the "
                                  "limit and its time follows the clock (DESIGN.md 4.1)"},
                "frac_alone": CORR_BYTES_PER_SAMPLE * float(nchan) * T / (float(np.mean(iso)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "energy_split_alone": energy_split() if tmpl.size > 512 else None,
                "note": "kernel_ms is measured over the timed region, where the timing-recovery kernel of the "
                        "previous step shares the chip; *_alone = same launch with nothing else running; copy_ceiling_GBs = "
                        "read + write rate of a plain float4 copy of 2 GiB on this box (hipEvents, no profiler)",
                "algorithmic_bytes_per_launch": CORR_BYTES_PER_SAMPLE * float(nchan) * T,
            },
            "parity": r["parity"],
            "detections_last_step": r["ndet"],
            "tag_overflow": r["tag_overflow"],
            "msk_status": int(st),
        }
        if ceil is not None:
            # what this box's HBM sustains by access shape (tools/ubench/hbm_ceiling.hip, 2 GiB buffers, hipEvents): the
            # correlator moves 8 B in + 8 B out per sample, so the copy figure is its practical ceiling
            line["roofline"]["achievable_GBs"] = ceil
            if ceil.get("best_copy_GBs"):
                line["roofline"]["frac_of_best_copy"] = achieved / ceil["best_copy_GBs"]
                line["roofline"]["frac_alone_of_best_copy"] = line["roofline"]["frac_alone"] * HBM_PEAK_GBS / ceil["best_copy_GBs"]
        if r.get("msk_ms"):
            mms = float(np.mean(r["msk_ms"]))
            mbytes = MSK_BYTES_PER_SAMPLE * float(nchan) * T
            line["roofline_msk"] = {
                "kernel": "k_msk", "bound": "latency (a recurrence per channel: 16384 dependent iteration pairs per step)",
                "bytes_per_sample": MSK_BYTES_PER_SAMPLE, "algorithmic_bytes_per_launch": mbytes, "kernel_ms": mms,
                "achieved": mbytes / (mms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": mbytes / (mms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "step_floor_ms": 4.75,
                "note": "the kernel that bounds the step at 4096 channels: ~290 ns per iteration pair whatever the channel "
                        "count (4.75 ms per 65536 samples alone on the chip: value can not exceed channels x 65536 / 4.75 ms)",
            }
        if h2d is not None:
            line["h2d"] = h2d
        if side and world == 1:
            line["config1_host_path"] = config1_host_path()
            if args.chain == "stock":
                # BASELINE config 5 as a side measurement (bench.py --chain wideband is the run of its own)
                w = bench_wideband(args, torch, device, as_dict=True, steps=20)
                line["config5_wideband"] = {k: w[k] for k in ("value", "unit", "ms_per_step", "steps", "pfb_ms", "demod_ms", "realtime_factor")}
                line["config5_wideband"]["workload"] = w["config"]["workload"]
        if c4 is not None:
            k4 = float(np.mean(c4["kern_ms"]))
            a4 = CORR_BYTES_PER_SAMPLE * 8192.0 * T / (k4 * 1e-3) / 1e9
            line["config4_per_gpu"] = {
                "what": "the same steps at BASELINE config 4's per-GPU shape, 8192 channels x %d samples (side measurement on this GPU)" % T,
                "ms_per_step": c4["el"] / args.steps * 1e3,
                "value": 8192.0 * T * args.steps / c4["el"] / 1e6, "unit": "complex MS/s",
                "corr_kernel_ms": k4, "corr_frac": a4 / HBM_PEAK_GBS,
                "msk_kernel_ms": float(np.mean(c4["msk_ms"])) if c4.get("msk_ms") else None,
                "msk_status": int(c4["st"]),
            }
        if extra is not None:
            line["corr_est_to_msk_only"] = {
                "chain": CHAIN_TEXT["core"],
                "value": total_samples / extra["el"] / 1e6,
                "unit": "complex MS/s",
                "ms_per_step": extra["el"] / args.steps * 1e3,
                "corr_kernel_ms": float(np.mean(extra["kern_ms"])),
                "detections_last_step": extra["ndet"],
                "msk_status": int(extra["st"]),
            }
        if nola is not None:
            line["no_lookahead_ms_per_step"] = nola["el"] / args.steps * 1e3
            # (there the phase walk runs in front of the pass on the same stream, not beside the correlator)
            line["no_lookahead_corr_kernel_ms"] = float(np.mean(nola["kern_ms"]))
            if nola.get("msk_ms"):
                line["no_lookahead_msk_kernel_ms"] = float(np.mean(nola["msk_ms"]))
        if coff is not None:
            line["corr_off_recovery_cus"] = {
                "what": "the same steps with aisx_corr_set_lds_claim(17408): no correlator workgroup fits beside a recovery workgroup "
                        "(the recovery runs less disturbed, the correlator on half of the CUs); not the default",
                "ms_per_step": coff["el"] / args.steps * 1e3, "value": nchan * world * float(T) * args.steps / coff["el"] / 1e6,
                "corr_kernel_ms": float(np.mean(coff["kern_ms"])),
                "msk_kernel_ms": float(np.mean(coff["msk_ms"])) if coff.get("msk_ms") else None,
                "msk_status": int(coff["st"]),
            }
        if tpm is not None:
            line["msk_time_parallel"] = {
                "what": "the same steps with aisx_msk_set_time_parallel(64 restart points, serial kernel as join, units <= 16384 items) "
                        "and set_max_noutput_items(256); off by default",
                "ms_per_step": tpm["el"] / args.steps * 1e3,
                "ms_per_step_serial_kernel_same_max_noutput_items": tpq["el"] / args.steps * 1e3,
                "corr_kernel_ms": float(np.mean(tpm["kern_ms"])),
                "restart_stats_last_step": tpm.get("restart_stats"),
                "parity": tpm["parity"],
                "msk_status": int(tpm["st"]),
            }
        if corr_only is not None:
            line["corr_only"] = corr_only
        if not args.no_cpu_baseline and world == 1:  # (the CPU path is timed beside the one-GPU run only)
            line["cpu_baseline"] = cpu_baseline(args.chain if args.chain != "corr" else "core", args.template, sps, T)
        # the side measurements once more under keys the driver's record keeps whole (it stores `config`, `roofline` and
        # `cpu_baseline` in full and only the NAMES of other top-level keys)
        sidek = {k: line[k] for k in ("config4_per_gpu", "corr_est_to_msk_only", "no_lookahead_ms_per_step", "no_lookahead_corr_kernel_ms",
                                       "no_lookahead_msk_kernel_ms", "config5_wideband", "config1_host_path", "h2d",
                                       "corr_off_recovery_cus") if k in line}
        if "msk_time_parallel" in line:
            sidek["msk_time_parallel"] = {k: line["msk_time_parallel"][k] for k in ("ms_per_step", "ms_per_step_serial_kernel_same_max_noutput_items")}
        if sidek:
            line["config"]["side"] = sidek
        if "roofline_msk" in line:
            line["roofline"]["msk"] = line["roofline_msk"]
        if "corr_only" in line:
            line["roofline"]["corr_only"] = line["corr_only"]
        line["roofline"]["parity_of_this_run"] = {k: line["parity"][k] for k in ("detections", "detections_matched_within_1", "mag_rtol_max",
                                                                                  "time_est_abs_max", "bursts_compared", "bursts_in_place",
                                                                                  "bursts_within_4") if line.get("parity") and k in line["parity"]}
        print(json.dumps(line))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
