"""feedforward AGC and square_and_fft_sync_cc / freqest kernel bodies under the
CPU lane model vs the oracle."""
import numpy as np
import pytest

import emul_py as emu
import oracle_py as orc
import synth


def test_emul_agc_bit_exact():
    rng = np.random.default_rng(2)
    nchan = 3
    lens = [5000, 1, 300, 2048, 4097, 9000]  # (9000: a second workgroup of four tiles per channel)
    total = sum(lens)
    x = (rng.normal(size=(nchan, total)) + 1j * rng.normal(size=(nchan, total))).astype(np.complex64)
    x[0, 1000:3000] = 0          # floor 1e-12 path
    x[1, 200] = np.nan           # NaN envelopes are ignored by std::max
    x[2] *= np.linspace(0.01, 30, total).astype(np.float32)
    for W in (512, 1, 37, 2048, 16, 24):
        e = emu.Agc(W, 2.0, nchan)
        o = [orc.Agc(W, 2.0) for _ in range(nchan)]
        k = 0
        for L in lens:
            out = e.work(x[:, k:k + L])
            for c in range(nchan):
                want = o[c].work(x[c, k:k + L])
                assert np.array_equal(out[c].view(np.uint32), want.view(np.uint32)), (W, L, c)
            k += L


def _check_freqsync(e, o, x, nchan):
    out, fh = e.process(x)
    nbad = 0
    for c in range(nchan):
        want, wfh = o[c].process(x[c])
        assert out.shape[1] == want.size and fh.shape[1] == wfh.size
        if np.array_equal(fh[c], wfh):
            assert np.array_equal(out[c].view(np.uint32), want.view(np.uint32))
        else:
            nbad += int(np.sum(fh[c] != wfh))
    return nbad, fh.size


def test_emul_freqsync_matches_oracle():
    nchan = 3
    fs = 38400.0
    lens = [4096, 1000, 24, 5000, 3 * 1024 + 7]
    total = sum(lens)
    xs = np.stack([synth.make_channel(300 + c, total, "P", 4, amp=0.5, cfo_max=500.0)[0] for c in range(nchan)])
    xs[1, 2048:5120] = 0  # all-zero vectors: the stale-maxpos quirk
    e = emu.FreqSync(fs, 9600.0, 1024, nchan)
    o = [orc.FreqSync(fs, 9600.0, 1024) for _ in range(nchan)]
    k = 0
    bad = tot = 0
    for L in lens:
        b, t = _check_freqsync(e, o, xs[:, k:k + L], nchan)
        bad += b
        tot += t
        k += L
    assert tot >= 3 * 12 and bad == 0


def test_emul_freqest_work_kat():
    e = emu.FreqSync(38400.0, 9600.0, 1024, 2)
    v = np.zeros((2, 3, 1024), dtype=np.complex64)
    v[0, 0, 394] = 5
    v[0, 0, 650] = 4j
    v[1, 1, 100] = 1
    v[1, 1, 356] = 1
    v[1, 1, 700] = 1
    v[1, 1, 956] = 1  # tie: the first maximum wins
    out = e.freqest_work(v)
    assert out[0].tolist() == [187.5, 187.5, 187.5]  # SURVEY 4.1 KAT + stale maxpos
    fe = orc.FreqEst.make(38400.0, 9600, 1024)
    assert np.array_equal(out[1], fe.work(v[1]))
    assert out[1][0] == -9600.0


@pytest.mark.parametrize("fftlen,sample_rate,data_rate", [(256, 38400.0, 9600), (1000, 48000.0, 9600), (2048, 50000.5, 9600),
                                                          (2, 48000.0, 9600), (64, 9600.0, 9600), (1024, 38400.0, 9600)])
def test_emul_freqest_any_vector_length(fftlen, sample_rate, data_rate):
    # freqest::work needs no transform of its own: every fftlen >= 2, incl. offset >= fftlen (nothing to search: 0 stays)
    from parity import freqest_cases

    v = freqest_cases(np.random.default_rng(fftlen), fftlen, sample_rate, data_rate)
    out = emu.freqest_any(v, v.shape[0], fftlen, sample_rate, data_rate)
    fe = orc.FreqEst.make(sample_rate, data_rate, fftlen)
    for c in range(v.shape[0]):
        assert np.array_equal(out[c].view(np.uint32), fe.work(v[c]).view(np.uint32)), (fftlen, c)


def test_emul_fused_front_end_is_freq_sync_then_agc():
    # aisx_freqsync_agc_process under the lane model: frequency estimates, the NCO phase walk on its
    # own (fs_walk_body) and the mixing inside the AGC's load stage (agc8_body) against the oracle's
    # freq_sync followed by its AGC, ragged calls (pending partial vectors, carried AGC history,
    # a call without a whole vector), an all-zero stretch (stale maxpos)
    nchan = 3
    lens = [4096, 1000, 24, 5000, 3 * 1024 + 7, 10, 9 * 1024]
    total = sum(lens)
    xs = np.stack([synth.make_channel(900 + c, total, "P", 4, amp=0.4, cfo_max=500.0)[0] for c in range(nchan)])
    xs[1, 2048:5120] = 0
    fs, agc = emu.FreqSync(38400.0, 9600.0, 1024, nchan), emu.Agc(512, 2.0, nchan)
    ofs = [orc.FreqSync(38400.0, 9600.0, 1024) for _ in range(nchan)]
    oag = [orc.Agc(512, 2.0) for _ in range(nchan)]
    k = nout = 0
    for L in lens:
        out, fh = emu.fs_agc_process(fs, agc, xs[:, k:k + L])
        for c in range(nchan):
            y, wfh = ofs[c].process(xs[c, k:k + L])
            want = oag[c].work(y) if y.size else y
            assert out.shape[1] == want.size and np.array_equal(fh[c], wfh)
            assert np.array_equal(out[c].view(np.uint32), want.view(np.uint32)), (L, c)
        nout += out.shape[1]
        k += L
    assert nout == (total // 1024) * 1024


def test_emul_streaming_agc_bit_exact():
    # agcw_body (k_agcw.h: a wave walks consecutive 512-item blocks, prefix / suffix maxima in
    # registers, wave scans) serves W = 512 with whole blocks of new items; every other call goes to
    # the tile kernels, and both carry the same history: ragged calls alternate between them
    rng = np.random.default_rng(5)
    nchan = 3
    lens = [512, 1024, 300, 8192 + 512, 7, 2 * 16 * 512 + 512, 2048, 5 * 512]  # (NB = 1; one run; two workgroups)
    total = sum(lens)
    x = (rng.normal(size=(nchan, total)) + 1j * rng.normal(size=(nchan, total))).astype(np.complex64)
    x[0, 600:3000] = 0            # the floor
    x[0, 20000:20600] = 0
    x[1, 200] = np.nan            # NaN envelopes are ignored by std::max
    x[1, 9000] = complex(np.nan, 1.0)
    x[1, 9600] = complex(2.0, np.inf)
    x[2] *= np.linspace(0.01, 30, total).astype(np.float32)
    x[2, 511] *= 1000             # a maximum in a block's last slot, another in a first slot
    x[2, 12800] *= 1000
    e = emu.Agc(512, 2.0, nchan)
    o = [orc.Agc(512, 2.0) for _ in range(nchan)]
    k = 0
    for L in lens:
        out = e.work(x[:, k:k + L])
        for c in range(nchan):
            want = o[c].work(x[c, k:k + L])
            assert np.array_equal(out[c].view(np.uint32), want.view(np.uint32)), (L, c)
        k += L


def test_nco_phase_to_fixed_is_float_to_fixed():
    # k_agcw.h folds phases in (-9.4, -pi) with one addition; everything else takes the general
    # statement: equal to gr::fxpt::float_to_fixed for every float the NCO's wrap can leave
    # (-3 pi .. pi) and beyond
    import ctypes as C
    pi = np.float32(np.pi)
    edges = []
    for centre in (-3 * np.pi, -9.4, -np.pi, 0.0, np.pi, -2 * np.pi, 1e-30, 12.0, -12.0, 100.0):
        c = np.float32(centre)
        v = np.array([c], dtype=np.float32).view(np.int32)[0]
        edges.append((np.arange(-4000, 4001, dtype=np.int64) + int(v)).astype(np.int32).view(np.float32))
    rng = np.random.default_rng(1)
    x = np.concatenate(edges + [rng.uniform(-3 * np.pi, np.pi, 2_000_000).astype(np.float32),
                                rng.uniform(-40, 40, 200_000).astype(np.float32),
                                np.array([0.0, -0.0, pi, -pi, 3 * pi, -3 * pi], dtype=np.float32)])
    x = np.ascontiguousarray(x[np.isfinite(x)])
    a = np.zeros(x.size, dtype=np.int32)
    b = np.zeros(x.size, dtype=np.int32)
    vp = C.c_void_p
    emu.lib().emu_nco_phase_to_fixed_n(vp(x.ctypes.data), vp(a.ctypes.data), C.c_long(x.size))
    orc.lib().orc_fxpt_float_to_fixed_n(vp(x.ctypes.data), vp(b.ctypes.data), C.c_long(x.size))
    assert np.array_equal(a, b)


def test_emul_streaming_front_end_signs_and_sizes():
    # the fused front end on agcw_body: channels with a strong carrier at a negative / positive /
    # near-Nyquist offset (phases below -pi for the whole call: the fold; increments of either sign),
    # calls that leave pending items, calls of several runs per channel, a call served by the tile
    # kernel in between (no whole vector)
    nchan = 4
    lens = [2048, 1500, 20 * 1024 + 300, 100, 1024 * 3 - 876, 24 * 512]
    total = sum(lens)
    rng = np.random.default_rng(77)
    fs_hz = 38400.0
    n = np.arange(total)
    xs = np.zeros((nchan, total), dtype=np.complex64)
    for c, f in enumerate((-3100.0, 2500.0, -9000.0, 40.0)):
        tone = np.exp(1j * (2 * np.pi * f / fs_hz * n + 0.3 * c))
        xs[c] = (0.5 * tone + 0.05 * (rng.normal(size=total) + 1j * rng.normal(size=total))).astype(np.complex64)
    xs[3, 5000:9000] = 0
    fs, agc = emu.FreqSync(fs_hz, 9600.0, 1024, nchan), emu.Agc(512, 2.0, nchan)
    ofs = [orc.FreqSync(fs_hz, 9600.0, 1024) for _ in range(nchan)]
    oag = [orc.Agc(512, 2.0) for _ in range(nchan)]
    k = 0
    seen = set()
    for L in lens:
        out, fh = emu.fs_agc_process(fs, agc, xs[:, k:k + L])
        for c in range(nchan):
            y, wfh = ofs[c].process(xs[c, k:k + L])
            want = oag[c].work(y) if y.size else y
            assert out.shape[1] == want.size and np.array_equal(fh[c], wfh)
            assert np.array_equal(out[c].view(np.uint32), want.view(np.uint32)), (L, c)
            seen.update(np.sign(wfh).tolist())
        k += L
    assert {-1.0, 1.0} <= seen


def test_emul_tile_front_end_still_exact():
    # the tile kernel (agc8_body) stays the path of every window but 512 and of calls without whole
    # blocks: the same ragged stream with the streaming kernel switched off
    nchan = 2
    lens = [4096, 1000, 24, 3 * 1024 + 7]
    total = sum(lens)
    xs = np.stack([synth.make_channel(950 + c, total, "P", 4, amp=0.4, cfo_max=500.0)[0] for c in range(nchan)])
    fs, agc = emu.FreqSync(38400.0, 9600.0, 1024, nchan), emu.Agc(512, 2.0, nchan)
    emu.lib().emu_agc_set_streaming(agc.h, 0)
    ofs = [orc.FreqSync(38400.0, 9600.0, 1024) for _ in range(nchan)]
    oag = [orc.Agc(512, 2.0) for _ in range(nchan)]
    k = 0
    for L in lens:
        out, fh = emu.fs_agc_process(fs, agc, xs[:, k:k + L])
        for c in range(nchan):
            y, wfh = ofs[c].process(xs[c, k:k + L])
            want = oag[c].work(y) if y.size else y
            assert np.array_equal(out[c].view(np.uint32), want.view(np.uint32)), (L, c)
        k += L
