"""-m gpu: feedforward AGC, square_and_fft_sync_cc / freqest and the stock chain
of python/ais_demod.py:56 on the GPU against the oracle."""
import numpy as np
import pytest

import oracle_py as orc
from parity import assert_aggregate_agreement, assert_decoded_bursts_identical, assert_tags_match

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ais():
    import torch

    assert torch.cuda.is_available()
    import ais_amd

    return ais_amd


def _dev(x):
    import torch

    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def test_agc_bit_exact(ais):
    rng = np.random.default_rng(2)
    nchan = 9
    lens = [5000, 1, 300, 2048, 40970]
    total = sum(lens)
    x = (rng.normal(size=(nchan, total)) + 1j * rng.normal(size=(nchan, total))).astype(np.complex64)
    x[0, 1000:3000] = 0
    x[1, 200] = np.nan
    x[2] *= np.linspace(0.01, 30, total).astype(np.float32)
    for W in (512, 1, 37, 2048, 16, 24):
        blk = ais.feedforward_agc_cc(W, 2.0, nchan=nchan, max_items=max(lens))
        o = [orc.Agc(W, 2.0) for _ in range(nchan)]
        k = 0
        for L in lens:
            out = blk.work(_dev(x[:, k:k + L])).cpu().numpy()
            for c in range(nchan):
                want = o[c].work(x[c, k:k + L])
                assert np.array_equal(out[c].view(np.uint32), want.view(np.uint32)), (W, L, c)
            k += L


def test_agc_other_references_and_huge_inputs(ais):
    # a reference that is not the stock 2, bit-exact against the oracle
    rng = np.random.default_rng(12)
    x = (rng.normal(size=(3, 5000)) + 1j * rng.normal(size=(3, 5000))).astype(np.complex64)
    out = ais.feedforward_agc_cc(512, 1.7, nchan=3, max_items=5000).work(_dev(x)).cpu().numpy()
    for c in range(3):
        assert np.array_equal(out[c].view(np.uint32), orc.Agc(512, 1.7).work(x[c]).view(np.uint32))
    # and window maxima far beyond anything a receiver sees (huge input)
    y = x.copy()
    y[0, 1000:1200] *= 1e32
    out = ais.feedforward_agc_cc(512, 2.0, nchan=3, max_items=5000).work(_dev(y)).cpu().numpy()
    for c in range(3):
        assert np.array_equal(out[c].view(np.uint32), orc.Agc(512, 2.0).work(y[c]).view(np.uint32))


def test_freqsync_matches_oracle(ais):
    import synth

    nchan = 70
    lens = [4096, 1000, 24, 5000, 30 * 1024 + 7]
    total = sum(lens)
    xs = np.stack([synth.make_channel(300 + c, total, "P", 4, amp=0.5, cfo_max=500.0)[0] for c in range(nchan)])
    xs[1, 2048:5120] = 0
    blk = ais.square_and_fft_sync_cc(38400.0, 9600.0, 1024, nchan=nchan, max_items=max(lens))
    o = [orc.FreqSync(38400.0, 9600.0, 1024) for _ in range(nchan)]
    k = 0
    nvec = nbad = 0
    for L in lens:
        out, fh = blk.work(_dev(xs[:, k:k + L]), want_fhat=True)
        out, fh = out.cpu().numpy(), fh.cpu().numpy()
        for c in range(nchan):
            want, wfh = o[c].process(xs[c, k:k + L])
            assert out.shape[1] == want.size and fh.shape[1] == wfh.size
            nvec += wfh.size
            if np.array_equal(fh[c], wfh):
                assert np.array_equal(out[c].view(np.uint32), want.view(np.uint32))
            else:
                # an estimate may only differ where two bin pairs tie to rounding; then
                # this channel's NCO phase departs from the oracle's for good
                nbad += 1
                o[c] = None
        o = [q if q is not None else orc.FreqSync(38400.0, 9600.0, 1024) for q in o]
        if nbad:
            break
        k += L
    assert nvec > 1000 and nbad == 0


def test_freqest_work_kat(ais):
    fe = ais.freqest(38400.0, 9600, 1024, nchan=2)
    v = np.zeros((2, 3, 1024), dtype=np.complex64)
    v[0, 0, 394] = 5
    v[0, 0, 650] = 4j
    v[1, 1, 100] = v[1, 1, 356] = v[1, 1, 700] = v[1, 1, 956] = 1
    out = fe.work(_dev(v.reshape(2, -1))).cpu().numpy()
    assert out[0].tolist() == [187.5, 187.5, 187.5]
    assert np.array_equal(out[1], orc.FreqEst.make(38400.0, 9600, 1024).work(v[1]))


@pytest.mark.parametrize("fftlen,sample_rate,data_rate", [(256, 38400.0, 9600), (1000, 48000.0, 9600), (2048, 50000.5, 9600),
                                                          (2, 48000.0, 9600), (64, 9600.0, 9600)])
def test_freqest_any_vector_length(ais, fftlen, sample_rate, data_rate):
    # the estimator block for every fftlen (include/ais/freqest.h:49): batched rows and the nchan = 1 host path, the
    # stale-maxpos rule (lib/freqest_impl.cc:68 vs :74), ties, offset >= fftlen; such a handle is refused by the freq_sync calls
    from parity import freqest_cases

    v = freqest_cases(np.random.default_rng(fftlen), fftlen, sample_rate, data_rate)
    nchan = v.shape[0]
    fe = ais.freqest(sample_rate, data_rate, fftlen, nchan=nchan)
    out = fe.work(_dev(v.reshape(nchan, -1))).cpu().numpy()
    o = orc.FreqEst.make(sample_rate, data_rate, fftlen)
    one = ais.freqest(sample_rate, data_rate, fftlen)
    for c in range(nchan):
        want = o.work(v[c])
        assert np.array_equal(out[c].view(np.uint32), want.view(np.uint32)), (fftlen, c)
        assert np.array_equal(one.work_host(v[c]).view(np.uint32), want.view(np.uint32))
    import ctypes as C

    from ais_amd import _lib

    L = _lib.lib()
    assert L.aisx_freqsync_is_estimator_only(fe._h) == 1
    buf = np.zeros(64, np.complex64)
    rc = L.aisx_freqsync_work_host(one._h, buf.ctypes.data_as(C.c_void_p), 4, buf.ctypes.data_as(C.c_void_p), 64, None, 0)
    assert rc == _lib.AISX_ERR_INVALID and b"aisx_freqest_create" in L.aisx_last_error()
    with pytest.raises(ValueError):  # (square_and_fft_sync_cc itself keeps its one vector length)
        ais.square_and_fft_sync_cc(sample_rate, float(data_rate), fftlen, nchan=1, max_items=4 * fftlen)


@pytest.mark.parametrize("family,nchan,T,steps", [("P", 24, 16384, 3), ("S", 24, 16384, 3)])
def test_stock_chain_bits_identical(ais, family, nchan, T, steps):
    # freq_sync -> agc -> corr_est -> msk -> NRZI bits, the connect order of
    # python/ais_demod.py:56, vs the oracle chain with the same step contract
    import synth

    sps = 4
    if family == "S":
        tmpl = ais.modulate_vector_bc(ais.gmsk_mod(sps, 0.4), [1, 1, 0, 0] * 7, [1])
    else:
        lv = [1 if b else -1 for b in synth.sync_bits("P")]
        tmpl = synth.gmsk_waveform(np.array(lv, float), sps)[: len(lv) * sps].astype(np.complex64)
    xs = np.stack([synth.make_channel(700 + c, T * steps, family, sps, amp=0.3, cfo_max=500.0)[0] for c in range(nchan)])
    opts = dict(samples_per_symbol=sps, bits_per_sec=9600.0, clockrec_gain=0.04, omega_relative_limit=0.01,
                fftlen=1024)
    dem = ais.ais_demod(opts, nchan=nchan, max_items=T, stages="stock", preamble_symbols=tmpl)
    ora = [orc.Demod(sps, tmpl, stages=3) for _ in range(nchan)]
    nbits = ntags = 0
    gbits = [[] for _ in range(nchan)]
    obits = [[] for _ in range(nchan)]
    for s in range(steps):
        chunk = xs[:, s * T:(s + 1) * T]
        r = dem.work(_dev(chunk))
        assert dem.clockrec.last_status() == 0
        prod = r["produced"].cpu().numpy()
        bits = r["bits"].cpu().numpy()
        tags = dem.preamble_detect.tags()
        for c in range(nchan):
            ob, _, ot = ora[c].step(chunk[c])
            ntags += assert_tags_match(tags[tags["chan"] == c], ot, exact_offsets=False)
            assert prod[c] == len(ob)
            gbits[c].append(bits[c, : prod[c]].copy())
            obits[c].append(ob)
            nbits += prod[c]
    ncmp = nburst = 0
    for c in range(nchan):
        _, infos = synth.make_channel(700 + c, T * steps, family, sps, amp=0.3, cfo_max=500.0)
        a, b = assert_decoded_bursts_identical(np.concatenate(gbits[c]), np.concatenate(obits[c]), infos)
        ncmp += a
        nburst += b
    print("stock chain %s: %d bits, %d detections within tolerance, %d decoded bursts bit-identical (of %d sent)"
          % (family, nbits, ntags, ncmp, nburst))
    assert_aggregate_agreement()
    # (achieved: P 2191 detections, 415 of 463 sent bursts; S 2030 detections, 237 of 295)
    assert ntags >= 2000 and ncmp >= int((0.88 if family == "P" else 0.79) * nburst)


def _match_detections_near_threshold(got, want, thr):
    """Detections of two correlators whose |corr|^2 agree to ~3e-7 relative: the lists are equal
    except where a peak sits within 2e-5 (relative) of the threshold, which one of them may then
    see and the other not (the magnitude tolerance is 1e-5).  Returns (matched, unmatched)."""
    from parity import MAG_RTOL, TIME_ATOL, tag_groups

    g, w = tag_groups(got), tag_groups(want)
    i = j = matched = unmatched = 0
    while i < len(g) or j < len(w):
        a = g[i] if i < len(g) else None
        b = w[j] if j < len(w) else None
        if a is not None and b is not None and abs(a["start"] - b["start"]) <= 1:
            assert abs(a["mag"] - b["mag"]) <= MAG_RTOL * abs(b["mag"]), (a, b)
            if a["start"] == b["start"]:
                assert abs(a["center"] - b["center"]) <= TIME_ATOL, (a, b)
            matched += 1
            i += 1
            j += 1
            continue
        lone = a if (b is None or (a is not None and a["start"] < b["start"])) else b
        assert abs(lone["mag"] - thr) <= 2e-5 * thr, ("unmatched detection away from the threshold", lone, thr)
        unmatched += 1
        if lone is a:
            i += 1
        else:
            j += 1
    return matched, unmatched


def test_stock_chain_full_length_steps(ais):
    # The benchmark's step (65536 samples, stock template, ~100 detections per channel and step:
    # the timing recovery's tag queue is refilled several times), 70 channels, two steps.
    #  (1) detections vs the oracle's own chain, within the tag tolerances (behind the AGC many
    #      peaks sit close to the threshold; one within 2e-5 of it may be seen by one correlator
    #      and not by the other);
    #  (2) the timing recovery against the oracle's block fed with the SAME items and tags (the
    #      GPU correlator's): symbols and counts bit for bit -- over a run this long a time_est
    #      that differs in its last place between the two correlators may move a symbol decision
    #      in the noise between bursts and with it the symbol count, so the chain-level bit gate
    #      of the shorter tests is applied burst by burst here:
    #  (3) every burst the oracle's chain decodes is in the GPU's bit stream within +-2 bits of
    #      the same place -- except, at most, one per detection that only one side saw (such a
    #      tag resets the timing loop in one chain and not in the other).
    import synth

    sps, nchan, T, steps = 4, 70, 65536, 2
    tmpl = ais.modulate_vector_bc(ais.gmsk_mod(sps, 0.4), [1, 1, 0, 0] * 7, [1])
    made = [synth.make_channel(1700 + c, T * steps, "S", sps, amp=0.3, cfo_max=500.0) for c in range(nchan)]
    xs = np.stack([m[0] for m in made])
    opts = dict(samples_per_symbol=sps, bits_per_sec=9600.0, clockrec_gain=0.04, omega_relative_limit=0.01,
                fftlen=1024)
    dem = ais.ais_demod(opts, nchan=nchan, max_items=T, stages="stock", preamble_symbols=tmpl)
    ora = [orc.Demod(sps, tmpl, stages=3) for _ in range(nchan)]
    omsk = [orc.MskStream(float(sps), 0.04, 0.01, 1) for _ in range(nchan)]
    obt = [orc.BitTail() for _ in range(nchan)]
    gbits = [[] for _ in range(nchan)]
    obits = [[] for _ in range(nchan)]
    ntags = nsym = nlone = 0
    lone_in = [0] * nchan
    for s in range(steps):
        chunk = xs[:, s * T:(s + 1) * T]
        y, _ = dem.freq_sync.work(_dev(chunk))
        y = dem.agc.work(y)
        yo, _ = dem.preamble_detect.work(y)
        tags = dem.preamble_detect.tags()
        r = dem.clockrec.work(yo, tags_from=dem.preamble_detect, want_syms=True)
        assert dem.clockrec.last_status() == 0
        prod = r["produced"].cpu().numpy()
        syms = r["syms"].cpu().numpy()
        bits = r["bits"].cpu().numpy()
        yo_h = yo.cpu().numpy()
        for c in range(nchan):
            tc = tags[tags["chan"] == c]
            ob, _, ot = ora[c].step(chunk[c])
            m, u = _match_detections_near_threshold(tc, ot, dem.preamble_detect.threshold())  # (1)
            ntags += m
            nlone += u
            lone_in[c] += u
            feed = np.zeros(len(tc), dtype=orc.TAG_DTYPE)
            feed["offset"], feed["value"], feed["key"] = tc["offset"], tc["value"], tc["key"]
            out, _, _, _ = omsk[c].step(yo_h[c], feed)
            assert prod[c] == len(out), (s, c)                                           # (2)
            assert np.array_equal(syms[c, : prod[c]].view(np.uint32), out.view(np.uint32)), (s, c)
            assert np.array_equal(bits[c, : prod[c]], obt[c].process(out)), (s, c)
            gbits[c].append(bits[c, : prod[c]].copy())
            obits[c].append(ob)
            nsym += prod[c]
    ncmp = nburst = nmiss = 0
    for c in range(nchan):
        g, o = np.concatenate(gbits[c]), np.concatenate(obits[c])
        assert abs(g.size - o.size) <= 2 + 2 * lone_in[c]
        miss = 0
        for inf in made[c][1]:                                                           # (3)
            pat = np.asarray(inf["data_bits"], dtype=np.uint8)
            nburst += 1
            for pos in synth.find_bits(o, pat):
                if any(np.array_equal(g[pos + d:pos + d + pat.size], pat) for d in range(-4, 5)):
                    ncmp += 1
                else:
                    miss += 1
        assert miss <= lone_in[c], (c, miss, lone_in[c])
        nmiss += miss
    print("full-length stock chain: %d symbols bit-exact given equal tags, %d detections within tolerance "
          "(%d at the threshold seen by one side only), %d decoded bursts found in place, %d not (of %d sent)"
          % (nsym, ntags, nlone, ncmp, nmiss, nburst))
    # (achieved: 15946 detections, 1 of them seen by one side only, 1894 of 2309 sent bursts decoded
    # by the oracle's chain, every one of them found in the GPU's stream)
    assert ntags > 15000 and nlone <= 3 and nmiss <= nlone and ncmp >= int(0.8 * nburst)


def test_fused_front_end_equals_the_two_blocks(ais):
    # aisx_freqsync_agc_process (estimates, NCO phase walk on its own, mixing inside the AGC's load
    # stage) against aisx_freqsync_process + aisx_agc_process on the GPU and against the oracle:
    # bit for bit, ragged calls, 70 channels (two walk waves, the second one ragged)
    import synth

    nchan = 70
    lens = [4096, 1000, 24, 5000, 30 * 1024 + 7, 10]
    total = sum(lens)
    xs = np.stack([synth.make_channel(1300 + c, total, "P", 4, amp=0.4, cfo_max=500.0)[0] for c in range(nchan)])
    xs[1, 2048:5120] = 0
    fs1 = ais.square_and_fft_sync_cc(38400.0, 9600.0, 1024, nchan=nchan, max_items=max(lens))
    ag1 = ais.feedforward_agc_cc(512, 2.0, nchan=nchan, max_items=max(lens) + 1024)
    fs2 = ais.square_and_fft_sync_cc(38400.0, 9600.0, 1024, nchan=nchan, max_items=max(lens))
    ag2 = ais.feedforward_agc_cc(512, 2.0, nchan=nchan, max_items=max(lens) + 1024)
    ofs = [orc.FreqSync(38400.0, 9600.0, 1024) for _ in range(8)]
    oag = [orc.Agc(512, 2.0) for _ in range(8)]
    k = nout = 0
    for L in lens:
        x = _dev(xs[:, k:k + L])
        a, fa = ais.freq_sync_agc(fs1, ag1, x, want_fhat=True)
        y, fb = fs2.work(x, want_fhat=True)
        b = ag2.work(y) if y.shape[1] else y
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), L
        assert np.array_equal(fa.cpu().numpy(), fb.cpu().numpy())
        for c in range(8):
            yo, _ = ofs[c].process(xs[c, k:k + L])
            want = oag[c].work(yo) if yo.size else yo
            assert np.array_equal(a[c].view(np.uint32), want.view(np.uint32)), (L, c)
        nout += a.shape[1]
        k += L
    assert nout == (total // 1024) * 1024
    with pytest.raises(ValueError):
        ais.freq_sync_agc(fs1, ais.feedforward_agc_cc(37, 2.0, nchan=nchan, max_items=max(lens) + 1024), _dev(xs[:, :100]))


def test_estimate_ahead_gives_the_same_results(ais):
    # aisx_freqsync_estimate_ahead: estimates + NCO phase walk of call k + 1 prepared on a second
    # stream while call k's sample pass runs; ragged calls (pending items written by the pass the
    # next estimate has to wait for), a prepared estimate that is dropped (other arguments)
    import torch
    import synth

    nchan = 70
    lens = [4096, 1000, 24, 5000, 9 * 1024 + 7, 10, 2048]
    total = sum(lens)
    xs = np.stack([synth.make_channel(1500 + c, total, "P", 4, amp=0.4, cfo_max=500.0)[0] for c in range(nchan)])
    mk = lambda: (ais.square_and_fft_sync_cc(38400.0, 9600.0, 1024, nchan=nchan, max_items=max(lens)),
                  ais.feedforward_agc_cc(512, 2.0, nchan=nchan, max_items=max(lens) + 1024))
    (fs1, ag1), (fs2, ag2) = mk(), mk()
    side = torch.cuda.Stream()
    chunks, k = [], 0
    for L in lens:
        chunks.append(_dev(xs[:, k:k + L]))
        k += L
    torch.cuda.synchronize()
    fs2.estimate_ahead(chunks[0], stream=side)
    for i, x in enumerate(chunks):
        a, fa = ais.freq_sync_agc(fs1, ag1, x, want_fhat=True)
        b, fb = ais.freq_sync_agc(fs2, ag2, x, want_fhat=True)
        if i + 1 < len(chunks):
            if i == 2:  # prepared for other arguments: must be dropped, not used
                fs2.estimate_ahead(chunks[0], stream=side)
            else:
                fs2.estimate_ahead(chunks[i + 1], stream=side)
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), i
        assert np.array_equal(fa.cpu().numpy(), fb.cpu().numpy())
    with pytest.raises(ValueError):
        fs2.estimate_ahead(chunks[0], stream=side)
        fs2.estimate_ahead(chunks[0], stream=side)  # a second one only when nothing is or will be pending


def test_estimate_two_calls_ahead(ais):
    # estimate_ahead(call k + 1) issued BEFORE freq_sync_agc(call k), as bench.py does: two
    # preparations wait at a time (two slots, three copies of the NCO phase); allowed behind calls
    # that leave no pending items; a call with other arguments drops both
    import torch
    import synth

    nchan = 70
    lens = [4096, 2048, 1024, 8192, 3072, 1000, 24, 2048]
    total = sum(lens)
    xs = np.stack([synth.make_channel(1700 + c, total, "P", 4, amp=0.4, cfo_max=500.0)[0] for c in range(nchan)])
    mk = lambda: (ais.square_and_fft_sync_cc(38400.0, 9600.0, 1024, nchan=nchan, max_items=max(lens)),
                  ais.feedforward_agc_cc(512, 2.0, nchan=nchan, max_items=max(lens) + 1024))
    (fs1, ag1), (fs2, ag2) = mk(), mk()
    side, walk = torch.cuda.Stream(), torch.cuda.Stream()
    chunks, k = [], 0
    for L in lens:
        chunks.append(_dev(xs[:, k:k + L]))
        k += L
    torch.cuda.synchronize()
    fs2.estimate_ahead(chunks[0], stream=side, walk_stream=walk)
    for i, x in enumerate(chunks):
        if i + 1 < len(chunks) and lens[i] % 1024 == 0 and sum(lens[:i]) % 1024 == 0:
            if i == 3:  # prepared for other arguments while another one waits: both are dropped
                fs2.estimate_ahead(chunks[1], stream=side, walk_stream=walk)
                b, fb = ais.freq_sync_agc(fs2, ag2, chunks[3][:, :4096].contiguous(), want_fhat=True)
                a, fa = ais.freq_sync_agc(fs1, ag1, chunks[3][:, :4096].contiguous(), want_fhat=True)
                assert np.array_equal(a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32))
                x = chunks[3][:, 4096:].contiguous()
            else:
                fs2.estimate_ahead(chunks[i + 1], stream=side, walk_stream=walk)  # two waiting now
                with pytest.raises(ValueError):
                    fs2.estimate_ahead(chunks[i + 1], stream=side, walk_stream=walk)  # a third
        a, fa = ais.freq_sync_agc(fs1, ag1, x, want_fhat=True)
        b, fb = ais.freq_sync_agc(fs2, ag2, x, want_fhat=True)
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), i
        assert np.array_equal(fa.cpu().numpy(), fb.cpu().numpy())
    # behind a call that leaves items pending no second estimate can be prepared
    fs3, ag3 = mk()
    fs3.estimate_ahead(chunks[5], stream=side)  # 1000 items
    with pytest.raises(ValueError):
        fs3.estimate_ahead(chunks[6], stream=side)


def test_ais_demod_fused_front_end_gives_the_same_bits(ais):
    # ais_demod(..., fused_front_end=True): the whole python/ais_demod.py:56 chain with freq_sync and
    # the AGC as one pass -- bits, symbol counts and tags equal to the chain with the two blocks
    import synth

    nchan, lens = 24, [8192, 5000, 12288]
    opts = dict(samples_per_symbol=4, bits_per_sec=9600.0, clockrec_gain=0.04, omega_relative_limit=0.01, fftlen=1024)
    xs = np.stack([synth.make_channel(2100 + c, sum(lens), "S", 4, amp=0.3, cfo_max=500.0)[0] for c in range(nchan)])
    a = ais.ais_demod(opts, nchan=nchan, max_items=max(lens))
    b = ais.ais_demod(opts, nchan=nchan, max_items=max(lens), fused_front_end=True)
    k = nbits = 0
    for L in lens:
        x = _dev(xs[:, k:k + L])
        ra, rb = a.work(x), b.work(x)
        pa, pb = ra["produced"].cpu().numpy(), rb["produced"].cpu().numpy()
        assert np.array_equal(pa, pb)
        ba, bb = ra["bits"].cpu().numpy(), rb["bits"].cpu().numpy()
        for c in range(nchan):
            assert np.array_equal(ba[c, :pa[c]], bb[c, :pb[c]])
        ta, tb = a.preamble_detect.tags(), b.preamble_detect.tags()
        assert ta.tobytes() == tb.tobytes()
        nbits += int(pa.sum())
        k += L
    assert nbits > nchan * sum(lens) / 4 * 0.9


def test_two_block_and_fused_calls_mixed_across_streams(ais):
    # aisx_freqsync_process (the two-block form) used to record no event, so a fused
    # call or an estimate prepared on ANOTHER stream right behind it could read the pending items,
    # slot 0's maxpos and the NCO phase while k_fs_mix was still writing them.  One handle, calls
    # alternating between the two forms and between two streams, no host synchronisation in between;
    # the output must be the oracle's chain of freq_sync -> agc, bit for bit.
    import torch
    import synth

    nchan = 70
    lens = [4096, 3000, 2048, 5000, 1024, 8192, 1000, 4096]
    xs = np.stack([synth.make_channel(5100 + c, sum(lens), "S", 4, amp=0.4, cfo_max=500.0)[0] for c in range(nchan)])
    fs = ais.square_and_fft_sync_cc(38400.0, 9600.0, 1024, nchan=nchan, max_items=max(lens))
    agc = ais.feedforward_agc_cc(512, 2.0, nchan=nchan, max_items=max(lens) + 1024)
    agc2 = ais.feedforward_agc_cc(512, 2.0, nchan=nchan, max_items=max(lens) + 1024)
    sA, sB, sW = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    chunks, k = [], 0
    for L in lens:
        chunks.append(_dev(xs[:, k:k + L]))
        k += L
    torch.cuda.synchronize()
    outs = []
    for i, x in enumerate(chunks):
        if i % 2 == 0:  # two-block form on stream A; its AGC on the same stream
            with torch.cuda.stream(sA):
                y, _ = fs.work(x, stream=sA)
                o = agc2.work(y, stream=sA) if y.shape[1] else y
                done = torch.cuda.Event()
                done.record(sA)
            if i + 1 < len(chunks) and i % 4 == 0:  # ... and the next call's estimates prepared at once on B / W
                fs.estimate_ahead(chunks[i + 1], stream=sB, walk_stream=sW)
        else:  # fused form on stream B (it consumes the preparation, or estimates for itself)
            with torch.cuda.stream(sB):
                sB.wait_event(done)  # (the AGC handles are two objects: order their histories by hand)
                o, _ = ais.freq_sync_agc(fs, agc, x, stream=sB)
        outs.append(o)
    torch.cuda.synchronize()
    # the two AGC objects each saw every other call: compare the freq_sync half through a fresh oracle
    # per form -- simplest exact statement: rerun the same call sequence serially on fresh handles
    fs_r = ais.square_and_fft_sync_cc(38400.0, 9600.0, 1024, nchan=nchan, max_items=max(lens))
    agc_r, agc2_r = (ais.feedforward_agc_cc(512, 2.0, nchan=nchan, max_items=max(lens) + 1024) for _ in range(2))
    for i, x in enumerate(chunks):
        if i % 2 == 0:
            y, _ = fs_r.work(x)
            r = agc2_r.work(y) if y.shape[1] else y
        else:
            r, _ = ais.freq_sync_agc(fs_r, agc_r, x)
        torch.cuda.synchronize()
        assert r.shape == outs[i].shape and torch.equal(r.view(torch.float32), outs[i].view(torch.float32)), i
    # and the freq_sync stream itself against the oracle (fhat decides everything downstream)
    o = [orc.FreqSync(38400.0, 9600.0, 1024) for _ in range(4)]
    fs_o = ais.square_and_fft_sync_cc(38400.0, 9600.0, 1024, nchan=nchan, max_items=max(lens))
    for i, x in enumerate(chunks):
        y, _ = fs_o.work(x)
        yh = y[:4].cpu().numpy()
        for c in range(4):
            want, _ = o[c].process(xs[c, sum(lens[:i]):sum(lens[:i + 1])])
            assert np.array_equal(yh[c].view(np.uint32), want.view(np.uint32)), (i, c)


@pytest.mark.gpu
def test_streaming_front_end_equals_tile_kernels_at_full_size(ais):
    # The stock window's calls run k_agcw.h (every wave walks its own run of 512-item blocks); the tile
    # kernels (k_agc.h) stay for every other window and length.  Same stream through both, fused
    # front end and the AGC alone: carriers at offsets of either sign (phases below -pi for whole
    # calls), silence (the floor), a NaN, calls that leave pending items, 65536-sample calls
    # (eight runs per channel), 300 channels; a few channels against the oracle.
    nchan = 300
    lens = [65536, 1000, 65536 + 24, 3 * 1024 - 1000, 512 * 33]
    total = sum(lens)
    rng = np.random.default_rng(9)
    n = np.arange(total)
    xs = (0.05 * (rng.normal(size=(nchan, total)) + 1j * rng.normal(size=(nchan, total)))).astype(np.complex64)
    offs = rng.uniform(-9000.0, 9000.0, nchan)
    for c in range(nchan):
        xs[c] += (0.5 * np.exp(1j * (2 * np.pi * offs[c] / 38400.0 * n + c))).astype(np.complex64)
    xs[3, 5000:70000] = 0
    xs[4, 777] = np.nan
    mk = lambda: (ais.square_and_fft_sync_cc(38400.0, 9600.0, 1024, nchan=nchan, max_items=max(lens)),
                  ais.feedforward_agc_cc(512, 2.0, nchan=nchan, max_items=max(lens) + 1024))
    fs1, ag1 = mk()
    fs2, ag2 = mk()
    ag2.set_streaming(False)
    pa1 = ais.feedforward_agc_cc(512, 2.0, nchan=nchan, max_items=max(lens))
    pa2 = ais.feedforward_agc_cc(512, 2.0, nchan=nchan, max_items=max(lens))
    pa2.set_streaming(False)
    nor = 6
    ofs = [orc.FreqSync(38400.0, 9600.0, 1024) for _ in range(nor)]
    oag = [orc.Agc(512, 2.0) for _ in range(nor)]
    k = 0
    for L in lens:
        x = _dev(xs[:, k:k + L])
        a, fa = ais.freq_sync_agc(fs1, ag1, x, want_fhat=True)
        b, fb = ais.freq_sync_agc(fs2, ag2, x, want_fhat=True)
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), L
        assert np.array_equal(fa.cpu().numpy(), fb.cpu().numpy())
        p, q = pa1.work(x).cpu().numpy(), pa2.work(x).cpu().numpy()
        assert np.array_equal(p.view(np.uint32), q.view(np.uint32)), L
        for c in range(nor):
            yo, _ = ofs[c].process(xs[c, k:k + L])
            want = oag[c].work(yo) if yo.size else yo
            assert np.array_equal(a[c].view(np.uint32), want.view(np.uint32)), (L, c)
        k += L


def test_streaming_agc_reciprocal_is_the_division(ais):
    # k_agcw.h: agcw_gain_fast -- with a power-of-two reference (the stock 2) the gain is one Newton step
    # from v_rcp_f32, scaled exactly, instead of the IEEE division sequence.  The hook sweeps EVERY float
    # max_env in [2^-100, 2^100] (1.68e9 values) on the device: it must agree with the division in
    # every bit, for the stock reference and two other powers of two.  Other references and maxima
    # beyond the range divide: checked against the oracle on the kernel itself.
    import ctypes as C
    from ais_amd import _lib

    for ref in (2.0, 1.0, 0.25):
        cnt, ex = C.c_ulonglong(123), C.c_float(0)
        _lib.check(_lib.lib().aisx_util_agc_rcp_mismatches(ref, C.byref(cnt), C.byref(ex)), "sweep")
        assert cnt.value == 0, (ref, cnt.value, ex.value)
    with pytest.raises(ValueError):
        _lib.check(_lib.lib().aisx_util_agc_rcp_mismatches(3.0, C.byref(cnt), C.byref(ex)), "sweep")
    rng = np.random.default_rng(12)
    x = (rng.normal(size=(3, 8192)) + 1j * rng.normal(size=(3, 8192))).astype(np.complex64)
    out = ais.feedforward_agc_cc(512, 1.7, nchan=3, max_items=8192).work(_dev(x)).cpu().numpy()
    for c in range(3):
        assert np.array_equal(out[c].view(np.uint32), orc.Agc(512, 1.7).work(x[c]).view(np.uint32))
    y = x.copy()
    y[:, 3000:3600] *= np.float32(1e33)  # maxima beyond 2^100 in some blocks, not in others
    y[1, 5000] = complex(3e38, 0)
    out = ais.feedforward_agc_cc(512, 2.0, nchan=3, max_items=8192).work(_dev(y)).cpu().numpy()
    for c in range(3):
        assert np.array_equal(out[c].view(np.uint32), orc.Agc(512, 2.0).work(y[c]).view(np.uint32))
    a = ais.feedforward_agc_cc(512, 2.0, nchan=3, max_items=8192)
    a.set_floor(1e-12)
    z = x.copy()
    z[0, 1000:4000] = 0
    out = a.work(_dev(z)).cpu().numpy()
    for c in range(3):
        assert np.array_equal(out[c].view(np.uint32), orc.Agc(512, 2.0, floor=1e-12).work(z[c]).view(np.uint32))
