#!/usr/bin/env python3
"""Regenerates the fixtures in this directory:  python tests/golden/make_golden.py

WHAT THESE ARE.  The reference ships no golden vectors, known-answer tests or
data files for this path (lib/qa_ais.cc:30-36 is an empty suite, python/ has no
qa_*.py), and its blocks cannot be built in this image (GNU Radio, VOLK and
Boost are absent), so nothing here comes from the reference or from a run of
it: parity stays "unpinned" in the sense of DESIGN.md section 5.  The fixtures
are REGRESSION vectors: seeded inputs and the outputs of this repository's own
CPU oracle (oracle/ais_oracle.c) at the commit that generated them, plus the
hand-derivable known answers SURVEY.md section 4.1 records.  They serve two
purposes:
  * -m "not gpu": the oracle must keep reproducing them bit for bit, so a
    change to the oracle cannot silently move the goal posts of the GPU tests;
  * -m gpu: the HIP path is compared with the stored outputs directly, without
    the oracle library in the loop.

Only data is stored (inputs, expected outputs, the transmitted payload bits);
the inputs come from tests/synth.py, this repository's own signal generator.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "gr-ais_amd"))

import oracle_py as orc  # noqa: E402
import synth  # noqa: E402  (tests/synth.py)

SPS = 4


def p_template(sps=SPS):
    lv = [1 if b else -1 for b in synth.sync_bits("P")]
    return synth.gmsk_waveform(np.array(lv, float), sps)[: len(lv) * sps].astype(np.complex64)


def corr_kat():
    """corr_est_cc, N = 20 template planted once with a +j rotation."""
    rng = np.random.default_rng(7)
    N = 20
    tmpl = np.exp(1j * rng.uniform(-np.pi, np.pi, N)).astype(np.complex64)
    x = np.zeros(360, dtype=np.complex64)
    x[100:120] = 1j * tmpl
    ce = orc.CorrEst(tmpl, 4.0, 1, 0.9)
    out, _, tags = ce.work(x)
    return dict(symbols=tmpl, sps=np.float32(4.0), mark_delay=np.int32(1), threshold=np.float32(0.9), x=x,
                out=out, tags=tags)


def corr_stream():
    """corr_est_cc, the 112-sample AIS training-sequence template over a noisy
    channel with bursts, three work() calls with carried history."""
    tmpl = p_template()
    lens = [4096, 777, 5000]
    x, infos = synth.make_channel(4242, sum(lens), "P", SPS, amp=1.0, cfo_max=40.0)
    ce = orc.CorrEst(tmpl, float(SPS), 1, 0.9)
    d = dict(symbols=tmpl, x=x, lens=np.array(lens, np.int32))
    k = 0
    for i, L in enumerate(lens):
        out, _, tags = ce.work(x[k:k + L])
        d["out%d" % i] = out
        d["tags%d" % i] = tags
        k += L
    return d


def msk_stream():
    """msk_timing_recovery_cc(4, 0.04, 0.01, osps) fed directly (no tags), ragged calls."""
    lens = [3000, 37, 2500, 1, 2654]
    x, _ = synth.make_channel(51, sum(lens), "P", SPS, amp=1.0, cfo_max=50.0)
    d = dict(x=x, lens=np.array(lens, np.int32), gain=np.float32(0.04), limit=np.float32(0.01))
    for osps in (1, 2):
        m = orc.MskStream(float(SPS), 0.04, 0.01, osps)
        bt = orc.BitTail()
        k = 0
        for i, L in enumerate(lens):
            out, err, mu, _ = m.step(x[k:k + L], np.zeros(0, orc.TAG_DTYPE), want_aux=True)
            d["osps%d_syms%d" % (osps, i)] = out
            d["osps%d_err%d" % (osps, i)] = err if err is not None else np.zeros(0, np.float32)
            d["osps%d_mu%d" % (osps, i)] = mu if mu is not None else np.zeros(0, np.float32)
            d["osps%d_bits%d" % (osps, i)] = bt.process(out)
            k += L
    return d


def agc_stream():
    """feedforward_agc_cc(W, 2.0) for W = 512 (stock) and 37, ragged calls with carried history;
    a silent stretch (the 1e-12 floor) and a NaN sample included."""
    lens = [1500, 1, 700, 1895]
    rng = np.random.default_rng(5)
    x = (rng.normal(size=sum(lens)) + 1j * rng.normal(size=sum(lens))).astype(np.complex64)
    x *= np.linspace(0.05, 20, x.size).astype(np.float32)
    x[600:1300] = 0
    x[2000] = np.nan
    d = dict(x=x, lens=np.array(lens, np.int32), reference=np.float32(2.0))
    for W in (512, 37):
        a = orc.Agc(W, 2.0)
        k = 0
        for i, L in enumerate(lens):
            d["w%d_out%d" % (W, i)] = a.work(x[k:k + L])
            k += L
    return d


def freqsync_stream():
    """square_and_fft_sync_cc(38400, 9600, 1024): NCO-corrected output and one frequency
    estimate per 1024-vector, ragged calls (a partial vector is carried), an all-zero vector
    (the stale-maxpos rule, lib/freqest_impl.cc:68 vs :74)."""
    lens = [4096, 1000, 24, 3072]
    x, _ = synth.make_channel(301, sum(lens), "P", SPS, amp=0.5, cfo_max=500.0)
    x[2048:3072] = 0
    f = orc.FreqSync(38400.0, 9600.0, 1024)
    d = dict(x=x, lens=np.array(lens, np.int32))
    k = 0
    for i, L in enumerate(lens):
        out, fh = f.process(x[k:k + L])
        d["out%d" % i] = out
        d["fhat%d" % i] = fh
        k += L
    return d


def chain(stages, seed0, amp, cfo):
    """python/ais_demod.py:56 for two channels, two steps of 8192 samples: stages = 0 is the
    hot path alone (corr_est -> msk -> bits), stages = 3 the whole connect order."""
    tmpl = p_template()
    nchan, T, steps = 2, 8192, 2
    d = dict(symbols=tmpl, nchan=np.int32(nchan), T=np.int32(T), steps=np.int32(steps))
    xs = []
    for c in range(nchan):
        x, infos = synth.make_channel(seed0 + c, T * steps, "P", SPS, amp=amp, cfo_max=cfo)
        xs.append(x)
        # the transmitted (stuffed) payload bit strings, concatenated, and their lengths
        pl = [np.asarray(i["data_bits"], np.uint8) for i in infos]
        d["payload%d" % c] = np.concatenate(pl) if pl else np.zeros(0, np.uint8)
        d["payload_len%d" % c] = np.array([p.size for p in pl], np.int32)
        dem = orc.Demod(SPS, tmpl, stages=stages)
        for s in range(steps):
            bits, _, tags = dem.step(x[s * T:(s + 1) * T])
            d["bits_c%d_s%d" % (c, s)] = bits
            d["tags_c%d_s%d" % (c, s)] = tags
    d["x"] = np.stack(xs)
    return d


def main():
    out = {
        "corr_kat.npz": corr_kat(),
        "corr_stream.npz": corr_stream(),
        "msk_stream.npz": msk_stream(),
        "agc_stream.npz": agc_stream(),
        "freqsync_stream.npz": freqsync_stream(),
        "chain_core.npz": chain(0, 900, 1.0, 15.0),
        "chain_stock.npz": chain(3, 700, 0.3, 500.0),
    }
    for name, d in out.items():
        path = os.path.join(HERE, name)
        np.savez_compressed(path, **d)
        print("%-18s %7d bytes  %d arrays" % (name, os.path.getsize(path), len(d)))


if __name__ == "__main__":
    main()
