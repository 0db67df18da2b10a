#!/usr/bin/env python3
"""bench.py -- complex MS/s through the demod hot path on MI355X, with the
corr_est_cc kernel's HBM-roofline fraction and a CPU baseline.

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the chain over one batch: CHANNELS_PER_GPU channels x
SAMPLES complex samples per GPU, already resident in HBM.  Weak scaling: every
rank owns its own channels, no data-path collective (channels are independent,
SURVEY.md section 8e); torch.distributed (RCCL) is used only for the timing
barrier and the max-over-ranks reduction.

Workload (BASELINE.json configs[2], the one the metric is quoted on): 4096
batched channels, 65536 samples each, sps = 4, stock template (N = 896, SURVEY
D4).  The default chain is the whole flowgraph of python/ais_demod.py:56:
freq_sync (square -> FFT -> freqest -> NCO mix) -> feedforward agc -> corr_est
-> msk_timing_recovery -> NRZI bit tail; `value` is its throughput.  The same
run also times `--chain core` (corr_est -> msk_timing only, the two blocks the
metric string names) and reports it under "corr_est_to_msk_only".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "gr-ais_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CORR_BYTES_PER_SAMPLE = 16  # 8 B read + 8 B delayed pass-through write


def make_template(family, sps):
    from ais_amd import gmsk_mod, modulate_vector_bc, synth

    if family == "S":
        return modulate_vector_bc(gmsk_mod(sps, 0.4), [1, 1, 0, 0] * 7, [1])
    lv = [1 if b else -1 for b in synth.sync_bits("P")]
    return synth.gmsk_waveform(np.array(lv, float), sps)[: len(lv) * sps].astype(np.complex64)


def make_input(nchan, T, family, sps, device, rank, stock):
    """Synthetic IQ resident on the device: `nuniq` CPU-generated channels
    (seeded) replicated with a per-channel carrier phase, plus
    per-sample device-generated noise so that no two channels are equal."""
    import torch
    from ais_amd import synth

    nuniq = 32
    amp = 0.3 if stock else 1.0
    cfo = 500.0 if stock else (15.0 if family == "P" else 3.0)
    base = np.stack([synth.make_channel(synth.SEED0 + 1000 * rank + c, T, family, sps, amp=amp, cfo_max=cfo,
                                        noise=False)[0] for c in range(nuniq)])
    g = torch.Generator(device=device)
    g.manual_seed(synth.SEED0 + rank)
    b = torch.as_tensor(base).to(device)
    reps = (nchan + nuniq - 1) // nuniq
    x = b.repeat(reps, 1)[:nchan].contiguous()
    ph = torch.rand(nchan, generator=g, device=device) * (2 * np.pi)
    x *= torch.polar(torch.ones_like(ph), ph).to(torch.complex64).view(-1, 1)
    sigma = amp * np.sqrt(sps / (10 ** (20.0 / 10.0)) / 2.0)  # Eb/N0 = 20 dB
    noise = torch.randn((nchan, T, 2), generator=g, device=device, dtype=torch.float32) * sigma
    x += torch.view_as_complex(noise)
    return x


def pmc_traffic(nchan, T, N):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC
    summary (collected in separate --pmc passes, see profiles/), if it was taken on
    this workload; None otherwise."""
    path = os.path.join(ROOT, "profiles", "r01_corr_main_pmc.json")
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None
    w = d.get("workload", {})
    if (w.get("channels"), w.get("samples"), w.get("template_len")) != (nchan, T, N):
        return None
    return d.get("hbm_bytes_per_launch")


def cpu_baseline(chain, family, sps, T, budget_s=12.0, max_ch=512):
    """The CPU oracle (a plain-C port of the reference's algorithm, single thread) on a bounded
    sample of the same workload: whole channels of T samples, one after the other, until about
    `budget_s` seconds of CPU work are done."""
    import oracle_py as orc
    from ais_amd import synth

    tmpl = make_template(family, sps)
    stock = chain == "stock"
    warm = orc.Demod(sps, tmpl, stages=3 if stock else 0)
    warm.step(synth.make_channel(synth.SEED0, 4096, family, sps)[0])  # warm the FFT plan cache
    spent, nch = 0.0, 0
    while spent < budget_s and nch < max_ch:
        x = synth.make_channel(synth.SEED0 + nch, T, family, sps, amp=0.3 if stock else 1.0,
                               cfo_max=500.0 if stock else 15.0)[0]
        dem = orc.Demod(sps, tmpl, stages=3 if stock else 0)
        t0 = time.perf_counter()
        dem.step(x)
        spent += time.perf_counter() - t0
        nch += 1
    return dict(value=nch * T / spent / 1e6, unit="complex MS/s", cores=1, kind="port",
                sample="%d channels x %d samples (%.1f s of CPU work), chain=%s, oracle/ais_oracle.c single thread"
                       % (nch, T, spent, chain))


def bench_wideband(args, torch, device):
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

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ev[0].record(); lanes = pfb.work(x); ev[1].record(); dem.work(lanes); ev[2].record()
    torch.cuda.synchronize()
    print(json.dumps({
        "metric": "wideband complex MS/s through polyphase channelizer -> 1024 demod lanes",
        "value": n * args.steps / el / 1e6, "unit": "complex MS/s (wideband input)", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "1 x %d wideband samples/step at 25 MS/s -> 1024 lanes x %d items (decim 512, 60227-tap "
                               "prototype) -> corr_est(N=1024)->msk" % (n, nfr)},
        "pfb_ms": ev[0].elapsed_time(ev[1]), "demod_ms": ev[1].elapsed_time(ev[2]),
        "realtime_factor": n * args.steps / el / 25e6}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--channels-per-gpu", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=65536)
    ap.add_argument("--template", choices=["S", "P"], default="S", help="S: stock 896-sample template; P: 112-sample preamble")
    ap.add_argument("--chain", choices=["core", "stock", "corr", "wideband"], default="stock",
                    help="stock = the whole ais_demod.py flowgraph (freq_sync with freqest, agc, corr_est, msk timing "
                         "recovery, NRZI tail); core = corr_est -> msk only; corr = corr_est only; wideband = BASELINE "
                         "config 5: one 25 MS/s stream -> 1024-lane polyphase channelizer -> core chain")
    ap.add_argument("--single-chain", action="store_true", help="do not add the corr_est->msk-only timing to a stock run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libaisx has no CPU path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ  # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)  # "nccl" is RCCL on ROCm

    import ais_amd

    if args.chain == "wideband":
        return bench_wideband(args, torch, device)
    sps, T, nchan = 4, args.samples, args.channels_per_gpu
    tmpl = make_template(args.template, sps)
    opts = dict(samples_per_symbol=sps, bits_per_sec=9600.0, clockrec_gain=0.04, omega_relative_limit=0.01, fftlen=1024)
    from ais_amd.shard import max_over_ranks

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(chain):
        """K timed steps of `chain` on this rank's channel shard; returns the wall time (max over
        ranks) and the correlator kernel's per-launch times inside the timed region."""
        stock = chain == "stock"
        x = make_input(nchan, T, args.template, sps, device, rank, stock)
        dem = ais_amd.ais_demod(opts, nchan=nchan, max_items=T, stages="stock" if stock else "core",
                                preamble_symbols=tmpl)
        corr = dem.preamble_detect
        corr.set_profiling(True)
        # preallocated inter-stage buffers, three of each in rotation: the timing recovery of
        # step k (latency-bound) runs on its own stream under the streaming stages of step k+1;
        # with a third buffer the streaming stages of step k+2 need not wait for it either
        # (the two sides take about the same time, and every wait of one for the other adds up)
        NBUF = 3
        y_corr = [torch.empty((nchan, T), dtype=torch.complex64, device=device) for _ in range(NBUF)]
        cap = dem.clockrec.out_capacity
        outs = [dict(syms=None, bits=torch.empty((nchan, cap), dtype=torch.uint8, device=device),
                     produced=torch.empty(nchan, dtype=torch.int32, device=device)) for _ in range(NBUF)]
        s_main, s_msk = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
        s_tail = torch.cuda.Stream(device=device)  # the bit tail of step k runs beside the recovery of step k+1
        dem.clockrec.set_tail_stream(s_tail)
        msk_done = [None] * NBUF
        state = dict(k=0)

        def step():
            k = state["k"]
            par = k % NBUF
            with torch.cuda.stream(s_main):
                if msk_done[par] is not None:
                    s_main.wait_event(msk_done[par])  # step k-3 released y_corr[par] and its tags
                y = x
                if stock:
                    y, _ = dem.freq_sync.work(y)
                    y = dem.agc.work(y)
                o, _ = corr.work(y, out=y_corr[par] if y.shape[1] == T else None)
                tags_ptrs = corr.tags_device()
                ready = torch.cuda.Event()
                ready.record(s_main)
            if chain != "corr":
                with torch.cuda.stream(s_msk):
                    s_msk.wait_event(ready)
                    dem.clockrec.work(o, tags_ptrs=tags_ptrs, outs=outs[par])
                    ev = torch.cuda.Event()
                    ev.record(s_msk)
                    msk_done[par] = ev
            state["k"] = k + 1

        for _ in range(args.warmup):
            step()
        barrier()
        corr.set_profiling(True)  # restart the event ring: the timed steps only
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        barrier()
        # per-launch duration of the dominant kernel over the timed region: hipEvents
        # recorded around it on its launch stream in every step, read back only now
        kern_ms = corr.kernel_ms_history()[-args.steps:]
        # the same kernel with the chip to itself (no timing-recovery kernel alongside)
        iso = []
        for _ in range(3):
            with torch.cuda.stream(s_main):
                corr.work(x if not stock else y_corr[0], out=y_corr[1])
            iso.append(corr.last_kernel_ms())
        torch.cuda.synchronize()
        el = max_over_ranks(el, device=device)
        res = dict(el=el, kern_ms=kern_ms, iso=iso, st=0, ndet=0)
        if rank == 0:
            res["st"] = dem.clockrec.last_status() if chain != "corr" else 0
            tags = corr.tags(allow_overflow=True)
            res["ndet"] = int((tags["key"] == 2).sum())
        del dem, x, y_corr, outs
        torch.cuda.empty_cache()
        return res

    CHAIN_TEXT = {"core": "corr_est->msk_timing+NRZI tail (no freq_sync / agc in front)",
                  "stock": "freq_sync(freqest)->agc->corr_est->msk_timing+NRZI tail (python/ais_demod.py:56)",
                  "corr": "corr_est only"}
    r = measure(args.chain)
    el, kern_ms, iso, st = r["el"], r["kern_ms"], r["iso"], r["st"]
    # the default run also times the two-block chain the metric string names, for reference
    extra = measure("core") if (args.chain == "stock" and not args.single_chain) else None

    if rank == 0:
        total_samples = float(nchan) * T * world * args.steps
        kms = float(np.mean(kern_ms))
        achieved = CORR_BYTES_PER_SAMPLE * float(nchan) * T / (kms * 1e-3) / 1e9
        line = {
            "metric": "complex MS/s through corr_est->msk_timing chain; corr_est %HBM roofline",
            "value": total_samples / el / 1e6,
            "unit": "complex MS/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%d batched channels/GPU x %d complex samples/step, sps=4, template N=%d (%s), chain=%s"
                % (nchan, T, tmpl.size, "stock ais_demod.py" if args.template == "S" else "28-symbol preamble",
                   CHAIN_TEXT[args.chain]),
                "channels_per_gpu": nchan,
                "samples_per_step": T,
                "template_len": int(tmpl.size),
                "chain": args.chain,
                "parallelism": "channel-sharded x%d, no collective" % world,
            },
            "roofline": {
                "kernel": "k_corr4_main" if tmpl.size > 512 else "k_corr_main",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(nchan, T, int(tmpl.size)),
                "traffic_source": "profiles/r01_corr_main_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
                "kernel_ms": kms,
                "kernel_ms_alone": float(np.mean(iso)),
                "frac_alone": CORR_BYTES_PER_SAMPLE * float(nchan) * T / (float(np.mean(iso)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "kernel_ms is measured over the timed region, where the timing-recovery kernel of the "
                        "previous step shares the chip; *_alone = same launch with nothing else running",
                "algorithmic_bytes_per_launch": CORR_BYTES_PER_SAMPLE * float(nchan) * T,
            },
            "detections_last_step": r["ndet"],
            "msk_status": int(st),
        }
        if extra is not None:
            line["corr_est_to_msk_only"] = {
                "chain": CHAIN_TEXT["core"],
                "value": total_samples / extra["el"] / 1e6,
                "unit": "complex MS/s",
                "ms_per_step": extra["el"] / args.steps * 1e3,
                "corr_kernel_ms": float(np.mean(extra["kern_ms"])),
                "detections_last_step": extra["ndet"],
            }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.chain if args.chain != "corr" else "core", args.template, sps, T)
        print(json.dumps(line))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
