"""-m gpu: the BASELINE.json configurations at their stated sizes, through the C ABI.

  config 2   256 batched channels x 65536 samples, corr_est only, N = 896 and N = 112
  config 3   4096 channels x 65536, the whole python/ais_demod.py:56 flowgraph (freq_sync + agc in
             front, stock template N = 896)
  config 4   the per-GPU shape of 65536 channels on 8 GPUs: 8192 channels x 65536
  SURVEY D3  corr_est at the other samples-per-symbol values the reference is run with
             (python/radio.py:49-57, python/ais.grc:79; isps at lib/corr_est_cc_impl.cc:193,270)

Oracle comparison on a subset of channels (the CPU oracle runs ~2.4 MS/s on the stock chain),
size-independent properties on all of them.
"""
import numpy as np
import pytest

import oracle_py as orc
from parity import assert_tags_match, compare_bursts, compare_detections

pytestmark = pytest.mark.gpu

SPS = 4
OPTS = dict(samples_per_symbol=SPS, bits_per_sec=9600.0, clockrec_gain=0.04, omega_relative_limit=0.01, fftlen=1024)


@pytest.fixture(scope="module")
def ais():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a visible MI355X"
    import ais_amd

    return ais_amd


def _dev(x):
    import torch

    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def _template(ais, family, sps=SPS):
    from ais_amd import synth

    if family == "S":
        return ais.modulate_vector_bc(ais.gmsk_mod(sps, 0.4), [1, 1, 0, 0] * 7, [1])
    lv = [1 if b else -1 for b in synth.sync_bits("P")]
    return synth.gmsk_waveform(np.array(lv, float), sps)[: len(lv) * sps].astype(np.complex64)


def _replicated(base, nchan):
    """channel c = base[c % len(base)] times a per-replica carrier phase (device side)."""
    import torch

    nu, T = base.shape
    reps = nchan // nu
    assert reps * nu == nchan
    x = _dev(base).repeat(reps, 1)
    rot = torch.exp(1j * torch.linspace(0, 6.0, reps, device="cuda")).to(torch.complex64)
    rot[0] = 1.0
    return (x.view(reps, nu, T) * rot.view(-1, 1, 1)).reshape(nchan, T).contiguous()


@pytest.mark.parametrize("family", ["S", "P"])
def test_config2_256_channels_corr_est_only(ais, family):
    import torch
    from ais_amd import synth

    nchan, T, nu = 256, 65536, 16
    tmpl = _template(ais, family)
    N = tmpl.size
    base = np.stack([synth.make_channel(2000 + c, 2 * T, family, SPS, amp=1.0, cfo_max=15.0 if family == "P" else 3.0)[0]
                     for c in range(nu)])
    blk = ais.corr_est_cc(tmpl, float(SPS), 1, 0.9, nchan=nchan, max_items=T, max_tags_per_chan=1024)
    ora = [orc.CorrEst(tmpl, float(SPS), 1, 0.9) for _ in range(nu)]
    ndet = 0
    prev = None
    for s in range(2):  # two calls: the second starts from a carried history
        chunk = base[:, s * T:(s + 1) * T]
        x = _replicated(chunk, nchan)
        out, _ = blk.work(x)
        torch.cuda.synchronize()
        # A2 on every channel: the output is the input delayed by N, bit for bit
        assert torch.equal(out[:, N:], x[:, : T - N])
        if prev is None:
            assert bool((out[:, :N] == 0).all())
        else:
            assert torch.equal(out[:, :N], prev[:, T - N:])
        prev = x
        tags = blk.tags()
        for c in range(nu):  # replica 0 (no rotation) against the oracle, offsets exact
            _, _, ot = ora[c].work(chunk[c])
            ndet += assert_tags_match(tags[tags["chan"] == c], ot)
        # |corr|^2 does not see the carrier phase: every replica finds the same detections (a peak
        # within rounding of the threshold may flip)
        cnt = np.bincount(tags["chan"][tags["key"] == 2], minlength=nchan).reshape(nchan // nu, nu)
        mism = int((cnt != cnt[0]).any(axis=1).sum())
        print("config 2 (%s) call %d: replica groups whose detection counts differ from group 0: %d of %d" % (family, s, mism, nchan // nu))
        assert mism <= 1  # (achieved: 0)
    print("config 2 (%s, N = %d): %d detections identical to the oracle on %d channels x 2 calls" % (family, N, ndet, nu))
    assert ndet > 20 * nu and blk.nitems_written() == 2 * T


def _stock_chain_against_oracle(ais, nchan, K, steps, seed0, base_noise_free=False):
    """The whole flowgraph on `nchan` channels x 65536 samples per step.  Channels 0..K-1 carry their
    own seeded waveform and are checked against the oracle with the gates of
    test_gpu_stages.py::test_stock_chain_full_length_steps; the others are replicas of them."""
    import torch
    from ais_amd import synth

    T = 65536
    tmpl = _template(ais, "S")
    made = [synth.make_channel(seed0 + c, T * steps, "S", SPS, amp=0.3, cfo_max=500.0) for c in range(K)]
    xs = np.stack([m[0] for m in made])
    dem = ais.ais_demod(OPTS, nchan=nchan, max_items=T, stages="stock", preamble_symbols=tmpl)
    thr = dem.preamble_detect.threshold()
    ora = [orc.Demod(SPS, tmpl, stages=3) for _ in range(K)]
    omsk = [orc.MskStream(float(SPS), 0.04, 0.01, 1) for _ in range(K)]
    obt = [orc.BitTail() for _ in range(K)]
    gbits = [[] for _ in range(K)]
    obits = [[] for _ in range(K)]
    tot = dict(detections=0, matched=0, lone=0, lone_near_threshold=0)
    mag = tim = 0.0
    nsym = 0
    for s in range(steps):
        chunk = xs[:, s * T:(s + 1) * T]
        x = _replicated(chunk, nchan)
        y, _ = dem.freq_sync.work(x)
        y = dem.agc.work(y)
        yo, _ = dem.preamble_detect.work(y)
        tags = dem.preamble_detect.tags()
        r = dem.clockrec.work(yo, tags_from=dem.preamble_detect, want_syms=True)
        assert dem.clockrec.last_status() == 0
        prod = r["produced"].cpu().numpy()
        # every channel: a symbol per ~sps samples, nothing truncated
        assert prod.min() > T // SPS - 64 and prod.max() < T // SPS + 64
        syms = r["syms"][:K].cpu().numpy()
        bits = r["bits"][:K].cpu().numpy()
        yo_h = yo[:K].cpu().numpy()
        # replica group 0 of every block of K channels is bit-identical to channels 0..K-1 only for
        # rotation 0; what every replica shares is the frequency estimate and the symbol count
        # (to within the few symbols a borderline detection moves)
        for c in range(K):
            tc = tags[tags["chan"] == c]
            ob, _, ot = ora[c].step(chunk[c])
            d = compare_detections(tc, ot, thr)
            for k in tot:
                tot[k] += d[k]
            mag, tim = max(mag, d["mag_rel_max"]), max(tim, d["time_est_abs_max"])
            feed = np.zeros(len(tc), dtype=orc.TAG_DTYPE)
            feed["offset"], feed["value"], feed["key"] = tc["offset"], tc["value"], tc["key"]
            out, _, _, _ = omsk[c].step(yo_h[c], feed)
            assert prod[c] == len(out), (s, c)
            assert np.array_equal(syms[c, : prod[c]].view(np.uint32), out.view(np.uint32)), (s, c)
            assert np.array_equal(bits[c, : prod[c]], obt[c].process(out)), (s, c)
            gbits[c].append(bits[c, : prod[c]].copy())
            obits[c].append(ob)
            nsym += prod[c]
        del x, y, yo, r
        torch.cuda.empty_cache()
    ncmp = same = near = 0
    for c in range(K):
        a, b, d = compare_bursts(np.concatenate(gbits[c]), np.concatenate(obits[c]), made[c][1])
        ncmp, same, near = ncmp + a, same + b, near + d
    print("stock chain at %d channels x %d steps: %d symbols bit-exact given equal tags on %d channels; %d detections, %d "
          "matched within +-1, %d seen by one side only (%d of them within 2e-5 of the threshold); mag rel max %.2e, time_est "
          "abs max %.2e; %d decoded bursts compared, %d identical in place, %d within +-4 bits"
          % (nchan, steps, nsym, K, tot["detections"], tot["matched"], tot["lone"], tot["lone_near_threshold"], mag, tim,
             ncmp, same, near))
    assert tot["matched"] > 50 * K * steps
    assert tot["lone"] == tot["lone_near_threshold"], "a detection away from the threshold is missing on one side"
    assert tot["lone"] <= max(2, tot["matched"] // 500)
    assert mag <= 1e-5 and tim <= 1e-4
    # (achieved at 4096 x 2 steps: 3685 of 3685 detections, 453 of 453 bursts identical in place)
    assert ncmp > 8 * K * steps and near >= ncmp - tot["lone"] and same >= ncmp - 2 * max(1, tot["lone"])
    return dem


def test_config3_4096_channels_stock_chain(ais):
    _stock_chain_against_oracle(ais, 4096, 16, 2, 3100)


def test_config4_per_gpu_shape_8192_channels(ais):
    # 65536 channels on 8 GPUs = 8192 per GPU (BASELINE config 4; the ranks share nothing)
    _stock_chain_against_oracle(ais, 8192, 8, 1, 3300)


def test_config4_rows_do_not_depend_on_their_position(ais):
    # 8192 channels whose upper half repeats the lower half: the same input row gives the same
    # bits and tags wherever it sits in the batch (workgroup, wave, lane)
    import torch
    from ais_amd import synth

    nchan, T, nu = 8192, 65536, 8
    tmpl = _template(ais, "S")
    base = np.stack([synth.make_channel(3500 + c, T, "S", SPS, amp=1.0, cfo_max=3.0)[0] for c in range(nu)])
    half = _replicated(base, nchan // 2)
    x = torch.cat([half, half], dim=0).contiguous()
    dem = ais.ais_demod(OPTS, nchan=nchan, max_items=T, stages="core", preamble_symbols=tmpl)
    r = dem.work(x)
    assert dem.clockrec.last_status() == 0
    prod = r["produced"]
    assert torch.equal(prod[: nchan // 2], prod[nchan // 2:])
    bits = r["bits"]
    m = int(prod.max())
    valid = torch.arange(m, device="cuda").view(1, -1) < prod[: nchan // 2].view(-1, 1)  # (rows are written up to `produced`)
    assert bool(((bits[: nchan // 2, :m] == bits[nchan // 2:, :m]) | ~valid).all())
    tags = dem.preamble_detect.tags()
    lo, hi = tags[tags["chan"] < nchan // 2], tags[tags["chan"] >= nchan // 2]
    assert len(lo) == len(hi) > 0
    assert np.array_equal(lo["offset"], hi["offset"]) and np.array_equal(lo["value"], hi["value"])
    assert np.array_equal(lo["chan"] + nchan // 2, hi["chan"])
    ob, _, ot = orc.Demod(SPS, tmpl, stages=0).step(base[0])
    assert_tags_match(tags[tags["chan"] == 0], ot, exact_offsets=False)
    assert int(prod[0]) == len(ob)


@pytest.mark.parametrize("sps_block,sps_signal", [(5.0, 5), (5.2083, 5), (3.0, 3), (4.4, 4)])
def test_corr_est_other_samples_per_symbol(ais, sps_block, sps_signal):
    # isps = (int)(sps + 0.5) steps the peak search (lib/corr_est_cc_impl.cc:193,270); the stock
    # receiver runs the block at 5.2083 sps against a 5 sps template (python/radio.py:49-57)
    from ais_amd import synth

    tmpl = _template(ais, "S", sps_signal)
    nchan, T = 12, 40000
    xs = np.stack([synth.make_channel(2600 + c, T, "S", sps_signal, amp=1.0, cfo_max=3.0)[0] for c in range(nchan)])
    blk = ais.corr_est_cc(tmpl, sps_block, 1, 0.9, nchan=nchan, max_items=T, max_tags_per_chan=2048)
    ora = [orc.CorrEst(tmpl, sps_block, 1, 0.9) for _ in range(nchan)]
    ndet = 0
    for lo, hi in ((0, 25000), (25000, 25001), (25001, T)):
        out, _ = blk.work(_dev(xs[:, lo:hi]))
        out = out.cpu().numpy()
        tags = blk.tags()
        for c in range(nchan):
            oo, _, ot = ora[c].work(xs[c, lo:hi])
            assert np.array_equal(out[c], oo)
            ndet += assert_tags_match(tags[tags["chan"] == c], ot)
    assert ndet > 2 * nchan


def test_set_symbols_with_a_new_length_follows_the_reference(ais):
    # lib/corr_est_cc_impl.cc:132-162: history, output multiple and mark_delay follow the new
    # length; taps are stored as given (no conjugate / reverse); d_thresh stays; the FFT filter
    # restarts from a zeroed tail while the delayed pass-through keeps the block's history
    from parity import planted, unit_template

    rng = np.random.default_rng(77)
    t1, t2, t3 = unit_template(rng, 200), unit_template(rng, 640), unit_template(rng, 96)
    nchan, n = 3, 6000
    blk = ais.corr_est_cc(t1, 4.0, 300, 0.5, nchan=nchan, max_items=n, max_tags_per_chan=1024)
    ora = [orc.CorrEst(t1, 4.0, 300, 0.5) for _ in range(nchan)]
    assert blk.mark_delay() == 199
    cur = t1
    for step, new in enumerate((None, t2, None, t3, t3)):
        if new is not None:
            blk.set_symbols(new)
            for o in ora:
                o.set_symbols(new)
            cur = new
            assert np.array_equal(blk.symbols(), new)
            assert blk.history() == new.size + 1 == ora[0].history
            assert blk.output_multiple() == ora[0].output_multiple
            assert blk.mark_delay() == ora[0].mark_delay and blk.threshold() == ora[0].threshold
        # set_symbols stores the taps as given, so a burst that matches them is conj(reverse(taps))
        sig = np.conj(cur[::-1]) if step > 0 and cur is not t1 else cur
        x = planted(rng, nchan, n, sig, [[500, 3000], [10], [n - sig.size - 5]], noise=0.05, amp=2.0)
        out, corr = blk.work(_dev(x), want_corr=True)
        out, corr = out.cpu().numpy(), corr.cpu().numpy()
        tags = blk.tags()
        for c in range(nchan):
            oo, oc, ot = ora[c].work(x[c], want_corr=True)
            assert np.array_equal(out[c], oo), (step, c)
            assert np.max(np.abs(corr[c] - oc)) / (np.max(np.abs(oc)) + 1e-30) < 2e-6, (step, c)
            assert_tags_match(tags[tags["chan"] == c], ot)
    with pytest.raises(ValueError):
        blk.set_symbols(unit_template(rng, 2500))


def test_agc_floor_both_values(ais):
    # [GR] feedforward_agc_cc: "float max_env = 1e-4; // avoid divide by zero, indirectly set max
    # gain" (live) vs the commented-out 1e-12: they differ only where a whole window stays below 1e-4
    rng = np.random.default_rng(9)
    nchan, n = 4, 9000
    x = (rng.normal(size=(nchan, n)) + 1j * rng.normal(size=(nchan, n))).astype(np.complex64)
    x[0, 2000:5000] *= 1e-6
    x[1, :] *= 3e-6
    x[2, 100:4000] = 0
    differs = 0
    for floor in (1e-4, 1e-12):
        blk = ais.feedforward_agc_cc(512, 2.0, nchan=nchan, max_items=n)
        if floor != 1e-4:
            blk.set_floor(floor)
        out = blk.work(_dev(x)).cpu().numpy()
        for c in range(nchan):
            want = orc.Agc(512, 2.0, floor=floor if floor != 1e-4 else None).work(x[c])
            assert np.array_equal(out[c].view(np.uint32), want.view(np.uint32)), (floor, c)
        differs += int(np.abs(out[1]).max() > 1.0)
    assert differs == 1  # the quiet channel is lifted to the reference level only with the 1e-12 floor
    with pytest.raises(ValueError):
        blk.set_floor(0.0)


def test_freqest_work_host_gnuradio_path(ais):
    # freqest::work as the scheduler calls it: host vectors in, one float per vector out; maxpos is
    # a local of the call (lib/freqest_impl.cc:68 vs :74): an all-zero first vector gives -9600 Hz
    fe = ais.freqest(38400.0, 9600, 1024, nchan=1)
    v = np.zeros((4, 1024), dtype=np.complex64)
    v[1, 394] = 5
    v[1, 650] = 4j
    v[3, 100] = v[3, 356] = 1
    got = fe.work_host(v)
    want = orc.FreqEst.make(38400.0, 9600, 1024).work(v)
    assert got.tolist() == want.tolist() == [-9600.0, 187.5, 187.5, want[3]]
    # a second call starts from maxpos = 0 again
    assert fe.work_host(v[2:3]).tolist() == [-9600.0]
    rng = np.random.default_rng(4)
    r = (rng.normal(size=(7, 1024)) + 1j * rng.normal(size=(7, 1024))).astype(np.complex64)
    assert np.array_equal(fe.work_host(r), orc.FreqEst.make(38400.0, 9600, 1024).work(r))
    with pytest.raises(ValueError):
        ais.freqest(38400.0, 9600, 1024, nchan=2).work_host(v)
