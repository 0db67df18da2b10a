"""corr_est_cc kernel bodies (gr-ais_amd/csrc/k_corr.h) run under the CPU lane
model and compared with the oracle.  Exercises the index arithmetic of the
2048-point register/LDS FFT, overlap-save tiling, history carry, threshold
bitmask and the tag resolver without a GPU."""
import numpy as np
import pytest

import emul_py as emu
import oracle_py as orc
from parity import assert_tags_match, planted, unit_template


def _oracle_run(tmpl, sps, chunks, mark_delay=1, thr=0.9, want_corr=True):
    o = orc.CorrEst(tmpl, sps, mark_delay, thr)
    outs, corrs, tags = [], [], []
    for x in chunks:
        a, b, t = o.work(x, want_corr=want_corr)
        outs.append(a)
        corrs.append(b)
        tags.append(t)
    return outs, corrs, tags, o


@pytest.mark.parametrize("N", [1, 20, 112, 512, 513, 896, 1024, 2048])
def test_emul_corr_dense_matches_oracle(N):
    rng = np.random.default_rng(100 + N)
    tmpl = unit_template(rng, N)
    n = 4500 if N < 512 else 3000
    pos = [[700, 2900], [5, n - N - 3], []]
    x = planted(rng, 3, n, tmpl, pos)
    e = emu.CorrEst(tmpl, 4.0, 1, 0.9, nchan=3)
    out, corr, tags, cnt, _ = e.work(x, want_corr=True)
    for c in range(3):
        oo, oc, ot, o = _oracle_run(tmpl, 4.0, [x[c]])
        assert e.threshold == o.threshold and e.output_multiple == o.output_multiple
        assert np.array_equal(e.symbols(), o.taps())
        assert np.array_equal(out[c], oo[0])
        scale = np.max(np.abs(oc[0])) + 1e-30
        assert np.max(np.abs(corr[c] - oc[0])) / scale < 2e-6
        assert_tags_match(tags[c], ot[0])
    if N > 1:
        assert sum(len(t) for t in tags) > 0


@pytest.mark.parametrize("N,nseg", [(112, 1), (112, 3), (896, 2)])
def test_emul_corr_sparse_streaming_segments(N, nseg):
    # sparse scratch (port 1 not connected), several successive calls with
    # state carried, different lengths incl. n < N, peaks on call boundaries
    rng = np.random.default_rng(7 * N + nseg)
    tmpl = unit_template(rng, N)
    lens = [3000, N // 2 + 1, 1, 2500, 4100]
    total = sum(lens)
    edges = np.cumsum(lens)
    pos = [[400, int(edges[0]) - N, int(edges[0]) - N + 1, int(edges[2]) + 10, int(edges[3]) - N // 2],
           [int(edges[0]) - N - 1, int(edges[3]) + 777]]
    xs = planted(rng, 2, total, tmpl, pos, noise=0.03)
    e = emu.CorrEst(tmpl, 4.0, 1, 0.9, nchan=2)
    o = [orc.CorrEst(tmpl, 4.0, 1, 0.9) for _ in range(2)]
    k = 0
    ndet = 0
    for L in lens:
        chunk = xs[:, k:k + L]
        out, _, tags, cnt, _ = e.work(chunk, want_corr=False, force_nseg=nseg)
        for c in range(2):
            oo, _, ot = o[c].work(chunk[c], want_corr=False)
            assert np.array_equal(out[c], oo)
            ndet += assert_tags_match(tags[c], ot)
        k += L
    assert ndet >= 5


def test_emul_corr_dense_detections_and_overflow():
    # threshold so low that (almost) every isps-th sample fires: the resolver's
    # serial chain, the climb across word boundaries and the tag-capacity overflow
    rng = np.random.default_rng(5)
    N = 20
    tmpl = unit_template(rng, N)
    n = 1500
    x = (rng.normal(size=(1, n)) + 1j * rng.normal(size=(1, n))).astype(np.complex64)
    e = emu.CorrEst(tmpl, 4.0, 3, 1e-4, nchan=1)
    out, _, tags, cnt, raw = e.work(x, want_corr=False, tag_cap=4 * n)
    o = orc.CorrEst(tmpl, 4.0, 3, 1e-4)
    _, _, ot = o.work(x[0])
    assert len(ot) > 4 * (n // 8)
    assert_tags_match(tags[0], ot)
    e2 = emu.CorrEst(tmpl, 4.0, 3, 1e-4, nchan=1)
    _, _, tags2, cnt2, _ = e2.work(x, want_corr=False, tag_cap=40)
    assert cnt2[0] == cnt[0] and len(tags2[0]) == 40
    assert np.array_equal(tags2[0], tags[0][:40])


def test_emul_corr_sps_rounding_and_mark_delay_clamp():
    rng = np.random.default_rng(9)
    N = 24
    tmpl = unit_template(rng, N)
    x = planted(rng, 1, 2000, tmpl, [[100, 900]], noise=0.2)
    for sps, md in [(5.2083, 1), (4.0, 1000), (2.4, 0)]:
        e = emu.CorrEst(tmpl, sps, md, 0.5, nchan=1)
        o = orc.CorrEst(tmpl, sps, md, 0.5)
        _, _, tags, _, _ = e.work(x)
        _, _, ot = o.work(x[0])
        assert_tags_match(tags[0], ot)


# (the product's builds, 5 and 6, at every length; their predecessors -- the experiments build's twins -- at three)
@pytest.mark.parametrize("mode,N", [(m, n) for m in (5, 6) for n in (896, 1120, 640, 513, 2047, 2048)] +
                         [(m, n) for m in (0, 1, 2, 3, 4) for n in (896, 513, 2047)])
def test_emul_corr_f4096_builds(mode, N):
    # the F = 4096 builds -- k_corr4k.h (0), k_corr4d.h with the template length folded in
    # where such a build exists (1) and with it at run time (2), k_corr4e.h (512 threads x 8 points,
    # four radix-8 passes) likewise (3, 4), k_corr4f.h (the same with
    # the next window in registers and one image) likewise (5, 6): several tiles per segment (the
    # window images alternate, the overlap is copied across, the next window arrives by "DMA"),
    # several segments, a ragged last tile, calls with carried history, a peak on a call edge
    emu.lib().emu_corr_set_dma(mode)
    try:
        rng = np.random.default_rng(31 * N + mode)
        tmpl = unit_template(rng, N)
        L = 4096 - N
        lens = [3 * L + 517, 40, 2 * L + 1]
        total = sum(lens)
        e0 = lens[0]
        pos = [[300, e0 - N, e0 + 30, total - N - 2], [L - 5, 2 * L + 100, e0 - N // 2]]
        xs = planted(rng, 2, total, tmpl, pos, noise=0.03)
        for nseg in (1, 2):
            e = emu.CorrEst(tmpl, 4.0, 1, 0.9, nchan=2)
            o = [orc.CorrEst(tmpl, 4.0, 1, 0.9) for _ in range(2)]
            k = ndet = 0
            for i, Ln in enumerate(lens):
                chunk = xs[:, k:k + Ln]
                dense = (i == 2)
                out, corr, tags, cnt, _ = e.work(chunk, want_corr=dense, force_nseg=nseg)
                for c in range(2):
                    oo, oc, ot = o[c].work(chunk[c], want_corr=dense)
                    assert np.array_equal(out[c], oo), (mode, N, nseg, i, c)
                    if dense:
                        assert np.max(np.abs(corr[c] - oc)) / (np.max(np.abs(oc)) + 1e-30) < 2e-6
                    ndet += assert_tags_match(tags[c], ot)
                k += Ln
            assert ndet >= 6
    finally:
        emu.lib().emu_corr_set_dma(1)



@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("N", [112, 140, 99, 1, 128, 129, 511, 512])
def test_emul_corr_f2048_builds(mode, N):
    # the three F = 2048 builds -- k_corr.h (0), k_corr2d.h with the template length folded in where
    # such a build exists (1) and with it at run time (2): several tiles per segment (the window
    # images alternate, what lies below the first DMA piece is copied across, the rest arrives by
    # "DMA"), several segments, a ragged last tile, calls with carried history, peaks on call and
    # tile edges, odd lengths (a DMA pair split between history and stream), N a whole piece
    emu.lib().emu_corr_set_dma(mode)
    try:
        rng = np.random.default_rng(17 * N + mode)
        tmpl = unit_template(rng, N)
        L = 2048 - N
        lens = [3 * L + 517, max(1, N // 3), 2 * L + 1, 40]
        total = sum(lens)
        e0 = lens[0]
        pos = [[300, e0 - N, e0 + 30, total - N - 2, L - N, L - N + 1], [L - 5, 2 * L + 100, e0 - N // 2, 0]]
        xs = planted(rng, 2, total, tmpl, pos, noise=0.03)
        for nseg in (1, 2):
            e = emu.CorrEst(tmpl, 4.0, 1, 0.9, nchan=2)
            o = [orc.CorrEst(tmpl, 4.0, 1, 0.9) for _ in range(2)]
            k = ndet = 0
            for i, Ln in enumerate(lens):
                chunk = xs[:, k:k + Ln]
                dense = (i == 2)
                out, corr, tags, cnt, _ = e.work(chunk, want_corr=dense, force_nseg=nseg)
                for c in range(2):
                    oo, oc, ot = o[c].work(chunk[c], want_corr=dense)
                    assert np.array_equal(out[c], oo), (mode, N, nseg, i, c)
                    if dense:
                        assert np.max(np.abs(corr[c] - oc)) / (np.max(np.abs(oc)) + 1e-30) < 2e-6
                    ndet += assert_tags_match(tags[c], ot)
                k += Ln
            assert ndet >= 6 or N == 1
    finally:
        emu.lib().emu_corr_set_dma(1)


@pytest.mark.parametrize("sps", [4.0, 9.6, 1.0])
def test_emul_corr_resolver_regions_at_block_boundaries(sps):
    # The resolver cuts a channel's hit bitmask into blocks of 4096 items and scans "regions" on different waves
    # (k_corr.h: corr_resolve_body).  Peaks planted so that the hit runs end just before, straddle, or start right
    # at multiples of 4096 -- clean and dirty block starts, a region of several blocks -- against the oracle's
    # sequential scan; a low threshold widens the runs around every peak.
    rng = np.random.default_rng(int(sps * 10))
    N = 20
    tmpl = unit_template(rng, N)
    n = 3 * 4096 + 700
    offs = list(range(-9, 7))
    pos = [[4096 - N + 1 + d, 8192 - N + 1 - d, 12288 - N + 1 + d // 2, 300 + 3 * d] for d in offs]
    x = planted(rng, len(offs), n, tmpl, pos, noise=0.02)
    e = emu.CorrEst(tmpl, sps, 1, 0.35, nchan=len(offs))
    _, _, tags, cnt, _ = e.work(x, want_corr=False)
    ndet = 0
    for c in range(len(offs)):
        o = orc.CorrEst(tmpl, sps, 1, 0.35)
        _, _, ot = o.work(x[c], want_corr=False)
        assert_tags_match(tags[c], ot)
        ndet += len(ot) // 4
    assert ndet >= 4 * len(offs)


def test_emul_corr_resolver_flood_falls_back_to_one_wave(monkeypatch):
    # more detections in a wave's regions than it can hold in LDS (rsv_det_cap: 64 with sixteen waves): the workgroup's first wave runs the
    # sequential scan; one quiet channel beside it takes the parallel path in the same launch
    monkeypatch.setenv("EMU_RSV_WAVES", "4")  # (four waves: 128 detections each; sixteen lane-model waves take a minute)
    rng = np.random.default_rng(77)
    N = 20
    tmpl = unit_template(rng, N)
    n = 4096 + 1200
    x = (rng.normal(size=(2, n)) + 1j * rng.normal(size=(2, n))).astype(np.complex64)
    x[0, 700:] *= 1e-3  # (the flood fills the head of block 0: ~170 detections in one wave's region)
    x[0, 4500:4500 + N] += tmpl
    x[1] *= 1e-3
    x[1, 4500:4500 + N] += tmpl
    e = emu.CorrEst(tmpl, 4.0, 3, 1e-4, nchan=2)
    _, _, tags, cnt, _ = e.work(x, want_corr=False, tag_cap=4 * n)
    for c in range(2):
        o = orc.CorrEst(tmpl, 4.0, 3, 1e-4)
        _, _, ot = o.work(x[c], want_corr=False)
        assert_tags_match(tags[c], ot)
    assert len(tags[0]) > 4 * 100 and 4 <= len(tags[1])
