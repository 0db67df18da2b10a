"""The committed fixtures of tests/golden/ (see make_golden.py there for what they
are and are not: regression vectors of this repository's oracle, the reference
has none to offer).

  -m "not gpu": the oracle reproduces every stored output bit for bit.
  -m gpu      : the HIP path through the C ABI against the stored outputs, with
                no oracle library in the loop (tolerances of tests/parity.py).
"""
import os

import numpy as np
import pytest

from parity import assert_tags_match

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SPS = 4


def _load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def _same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


# ---------------------------------------------------------------- CPU: oracle pinned to the fixtures

def test_oracle_reproduces_corr_fixtures():
    import oracle_py as orc

    g = _load("corr_kat")
    ce = orc.CorrEst(g["symbols"], float(g["sps"]), int(g["mark_delay"]), float(g["threshold"]))
    out, _, tags = ce.work(g["x"])
    assert _same_bits(out, g["out"]) and _same_bits(tags, g["tags"])
    # the hand-derivable part: one detection, marked at 119/120, |corr|^2 = 400
    assert [int(t["offset"]) for t in g["tags"]] == [119, 120, 120, 120]
    assert abs(float(g["tags"][0]["value"]) - 400) < 1e-2

    g = _load("corr_stream")
    ce = orc.CorrEst(g["symbols"], float(SPS), 1, 0.9)
    k = 0
    for i, L in enumerate(g["lens"]):
        out, _, tags = ce.work(g["x"][k:k + L])
        assert _same_bits(out, g["out%d" % i]) and _same_bits(tags, g["tags%d" % i]), i
        k += L


def test_oracle_reproduces_msk_fixture():
    import oracle_py as orc

    g = _load("msk_stream")
    for osps in (1, 2):
        m = orc.MskStream(float(SPS), float(g["gain"]), float(g["limit"]), osps)
        bt = orc.BitTail()
        k = n = 0
        for i, L in enumerate(g["lens"]):
            out, err, mu, _ = m.step(g["x"][k:k + L], np.zeros(0, orc.TAG_DTYPE), want_aux=True)
            assert _same_bits(out, g["osps%d_syms%d" % (osps, i)]), (osps, i)
            if out.size:
                assert _same_bits(err, g["osps%d_err%d" % (osps, i)]) and _same_bits(mu, g["osps%d_mu%d" % (osps, i)])
            assert _same_bits(bt.process(out), g["osps%d_bits%d" % (osps, i)])
            k += L
            n += out.size
        assert n > osps * k // SPS - 16


def test_oracle_reproduces_agc_and_freqsync_fixtures():
    import oracle_py as orc

    g = _load("agc_stream")
    for W in (512, 37):
        a = orc.Agc(W, float(g["reference"]))
        k = 0
        for i, L in enumerate(g["lens"]):
            assert _same_bits(a.work(g["x"][k:k + L]), g["w%d_out%d" % (W, i)]), (W, i)
            k += L
    g = _load("freqsync_stream")
    f = orc.FreqSync(38400.0, 9600.0, 1024)
    k = nvec = 0
    for i, L in enumerate(g["lens"]):
        out, fh = f.process(g["x"][k:k + L])
        assert _same_bits(out, g["out%d" % i]) and _same_bits(fh, g["fhat%d" % i]), i
        k += L
        nvec += fh.size
    assert nvec == 8


@pytest.mark.parametrize("name,stages", [("chain_core", 0), ("chain_stock", 3)])
def test_oracle_reproduces_chain_fixtures(name, stages):
    import oracle_py as orc

    g = _load(name)
    T = int(g["T"])
    for c in range(int(g["nchan"])):
        dem = orc.Demod(SPS, g["symbols"], stages=stages)
        for s in range(int(g["steps"])):
            bits, _, tags = dem.step(g["x"][c, s * T:(s + 1) * T])
            assert _same_bits(bits, g["bits_c%d_s%d" % (c, s)]), (c, s)
            assert _same_bits(tags, g["tags_c%d_s%d" % (c, s)]), (c, s)


def test_chain_fixtures_contain_the_transmitted_payloads():
    # the stored bit streams are worth comparing against: most of the bursts that were sent
    # come out of them bit for bit
    import synth

    for name in ("chain_core", "chain_stock"):
        g = _load(name)
        found = sent = 0
        for c in range(int(g["nchan"])):
            bits = np.concatenate([g["bits_c%d_s%d" % (c, s)] for s in range(int(g["steps"]))])
            k = 0
            for L in g["payload_len%d" % c]:
                found += len(synth.find_bits(bits, g["payload%d" % c][k:k + L])) > 0
                sent += 1
                k += L
        assert sent >= 10 and found >= sent // 2, (name, found, sent)


# ---------------------------------------------------------------- GPU: HIP path against the fixtures

@pytest.fixture(scope="module")
def ais():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a visible MI355X"
    import ais_amd

    return ais_amd


def _dev(x):
    import torch

    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


@pytest.mark.gpu
def test_gpu_corr_against_fixtures(ais):
    g = _load("corr_kat")
    blk = ais.corr_est_cc(g["symbols"], float(g["sps"]), int(g["mark_delay"]), float(g["threshold"]), nchan=1,
                          max_items=g["x"].size)
    out, _ = blk.work(_dev(g["x"][None, :]))
    assert _same_bits(out.cpu().numpy()[0], g["out"])
    assert assert_tags_match(blk.tags(), g["tags"]) == 1

    g = _load("corr_stream")
    blk = ais.corr_est_cc(g["symbols"], float(SPS), 1, 0.9, nchan=1, max_items=int(max(g["lens"])))
    k = ndet = 0
    for i, L in enumerate(g["lens"]):
        out, _ = blk.work(_dev(g["x"][None, k:k + L]))
        assert _same_bits(out.cpu().numpy()[0], g["out%d" % i])
        ndet += assert_tags_match(blk.tags(), g["tags%d" % i])
        k += L
    assert ndet >= 4


@pytest.mark.gpu
@pytest.mark.parametrize("osps", [1, 2])
def test_gpu_msk_against_fixture(ais, osps):
    g = _load("msk_stream")
    blk = ais.msk_timing_recovery_cc(float(SPS), float(g["gain"]), float(g["limit"]), osps, nchan=1,
                                     max_items=int(max(g["lens"])))
    k = 0
    for i, L in enumerate(g["lens"]):
        r = blk.work(_dev(g["x"][None, k:k + L]), want_aux=True)
        assert blk.last_status() == 0
        p = int(r["produced"].cpu().numpy()[0])
        want = g["osps%d_syms%d" % (osps, i)]
        assert p == want.size
        assert _same_bits(r["syms"].cpu().numpy()[0, :p], want), i
        assert _same_bits(r["err"].cpu().numpy()[0, :p], g["osps%d_err%d" % (osps, i)])
        assert _same_bits(r["mu"].cpu().numpy()[0, :p], g["osps%d_mu%d" % (osps, i)])
        assert _same_bits(r["bits"].cpu().numpy()[0, :p], g["osps%d_bits%d" % (osps, i)])
        k += L


@pytest.mark.gpu
def test_gpu_agc_and_freqsync_against_fixtures(ais):
    g = _load("agc_stream")
    for W in (512, 37):
        blk = ais.feedforward_agc_cc(W, float(g["reference"]), nchan=1, max_items=int(max(g["lens"])))
        k = 0
        for i, L in enumerate(g["lens"]):
            out = blk.work(_dev(g["x"][None, k:k + L])).cpu().numpy()[0]
            assert _same_bits(out, g["w%d_out%d" % (W, i)]), (W, i)
            k += L
    g = _load("freqsync_stream")
    blk = ais.square_and_fft_sync_cc(38400.0, 9600.0, 1024, nchan=1, max_items=int(max(g["lens"])))
    k = 0
    for i, L in enumerate(g["lens"]):
        out, fh = blk.work(_dev(g["x"][None, k:k + L]), want_fhat=True)
        # (the estimate of a vector could only differ where two bin pairs tie to rounding: not in this fixture)
        assert _same_bits(fh.cpu().numpy()[0], g["fhat%d" % i]), i
        assert _same_bits(out.cpu().numpy()[0], g["out%d" % i]), i
        k += L


@pytest.mark.gpu
@pytest.mark.parametrize("name,stages", [("chain_core", "core"), ("chain_stock", "stock")])
def test_gpu_chain_against_fixtures(ais, name, stages):
    import synth

    g = _load(name)
    nchan, T, steps = int(g["nchan"]), int(g["T"]), int(g["steps"])
    opts = dict(samples_per_symbol=SPS, bits_per_sec=9600.0, clockrec_gain=0.04, omega_relative_limit=0.01,
                fftlen=1024)
    dem = ais.ais_demod(opts, nchan=nchan, max_items=T, stages=stages, preamble_symbols=g["symbols"])
    got = [[] for _ in range(nchan)]
    ndet = 0
    for s in range(steps):
        r = dem.work(_dev(g["x"][:, s * T:(s + 1) * T]))
        assert dem.clockrec.last_status() == 0
        prod = r["produced"].cpu().numpy()
        bits = r["bits"].cpu().numpy()
        tags = dem.preamble_detect.tags()
        for c in range(nchan):
            want = g["bits_c%d_s%d" % (c, s)]
            assert prod[c] == want.size
            ndet += assert_tags_match(tags[tags["chan"] == c], g["tags_c%d_s%d" % (c, s)], exact_offsets=False)
            got[c].append(bits[c, : prod[c]].copy())
    ncmp = 0
    for c in range(nchan):
        gb = np.concatenate(got[c])
        wb = np.concatenate([g["bits_c%d_s%d" % (c, s)] for s in range(steps)])
        # every transmitted payload the stored stream holds is in the GPU's stream, same place
        k = 0
        for L in g["payload_len%d" % c]:
            pat = g["payload%d" % c][k:k + L]
            for pos in synth.find_bits(wb, pat):
                assert np.array_equal(gb[pos:pos + L], pat), (c, pos)
                ncmp += 1
            k += L
        assert float(np.mean(gb == wb)) >= 0.97
    assert ndet >= 4 and ncmp >= 5
