"""msk_timing_recovery_cc kernel body (gr-ais_amd/csrc/k_msk.h) under the CPU
lane model vs the oracle: every output must be BIT-identical."""
import numpy as np
import pytest

import emul_py as emu
import oracle_py as orc
import synth


def _signal(seed, n, sps=4, family="P"):
    x, infos = synth.make_channel(seed, n, family, sps, amp=1.0, cfo_max=50.0)
    return x, infos


def _rand_tags(rng, n, count, chan=0):
    offs = np.sort(rng.choice(np.arange(10, n - 10), size=count, replace=False))
    tags = np.zeros(count, dtype=emu.TAG_DTYPE)
    tags["offset"] = offs
    tags["value"] = rng.uniform(-0.9, 0.9, count)
    tags["key"] = 2
    tags["chan"] = chan
    return tags


@pytest.mark.parametrize("sps,osps,lpw", [(4.0, 1, 64), (4.0, 2, 64), (5.2083, 1, 32), (5.0, 1, 16), (4.0, 1, 8),
                                          (4.0, 2, 4), (4.0, 1, 4)])
def test_emul_msk_stream_bit_exact(sps, osps, lpw):
    rng = np.random.default_rng(int(sps * 10) + osps)
    # > 64 channels so that both channels of a lane are live in the 2-channels-per-lane kernel
    nchan, lens = (67 if (sps, osps, lpw) in ((4.0, 1, 64), (4.0, 1, 8)) else (21 if lpw == 4 else 3)), [1500, 37, 900, 1, 700]
    total = sum(lens)
    xs = np.stack([_signal(50 + c, total, 4)[0] for c in range(nchan)])
    e = emu.MskStream(sps, 0.04, 0.01, osps, nchan=nchan, lpw=lpw)  # channels per wave
    o = [orc.MskStream(sps, 0.04, 0.01, osps) for _ in range(nchan)]
    bt = [orc.BitTail() for _ in range(nchan)]
    # tags: a mix of plausible time_est tags, NaN, other keys, clustered offsets
    all_tags = []
    for c in range(nchan):
        t = _rand_tags(rng, total, 30, c)
        t["value"][5] = np.nan
        t["key"][7] = 1
        t["offset"][20] = t["offset"][19] + 1
        t["offset"][21] = t["offset"][19] + 2
        t = t[np.argsort(t["offset"], kind="stable")]
        all_tags.append(t)
    k = 0
    nsym = 0
    for L in lens:
        chunk = xs[:, k:k + L]
        cap = 32
        tg = np.zeros((nchan, cap), dtype=emu.TAG_DTYPE)
        cnt = np.zeros(nchan, np.int32)
        new = []
        for c in range(nchan):
            sel = all_tags[c][(all_tags[c]["offset"] >= k) & (all_tags[c]["offset"] < k + L)]
            tg[c, : len(sel)] = sel
            cnt[c] = len(sel)
            new.append(sel)
        r = e.step(chunk, tg, cnt, want_aux=True)
        assert r["status"] == 0
        for c in range(nchan):
            ot = np.zeros(len(new[c]), dtype=orc.TAG_DTYPE)
            ot["offset"], ot["value"], ot["key"] = new[c]["offset"], new[c]["value"], new[c]["key"]
            out, o2, o3, cons = o[c].step(chunk[c], ot, want_aux=True)
            p = r["produced"][c]
            assert p == len(out) and r["consumed"][c] == cons
            assert np.array_equal(r["syms"][c, :p].view(np.uint32), out.view(np.uint32))
            if p:
                assert np.array_equal(r["err"][c, :p].view(np.uint32), o2.view(np.uint32))
                assert np.array_equal(r["mu"][c, :p].view(np.uint32), o3.view(np.uint32))
            assert np.array_equal(r["bits"][c, :p], bt[c].process(out))
            nsym += p
        k += L
    assert nsym > nchan * total / sps * osps * 0.95


def test_emul_msk_general_work_gr_mode():
    # the GNU Radio path: explicit ninput/noutput, caller re-presents unconsumed items
    rng = np.random.default_rng(3)
    x, _ = _signal(77, 6000)
    buf = np.concatenate([np.zeros(1, np.complex64), x])
    tags = _rand_tags(rng, 6000, 20)
    ot = np.zeros(len(tags), dtype=orc.TAG_DTYPE)
    ot["offset"], ot["value"], ot["key"] = tags["offset"], tags["value"], tags["key"]
    e = emu.MskStream(4.0, 0.04, 0.01, 1, nchan=1)
    o = orc.Msk(4.0, 0.04, 0.01, 1)
    read = 0
    for nout in [300, 100, 1, 400, 33]:
        ninput = o.forecast(nout) + int(rng.integers(0, 40))
        if read + ninput + 1 > x.size:
            break
        a = e.general_work(nout, ninput, buf, 1 + read, tags, read)
        b = o.general_work(nout, ninput, buf, 1 + read, ot, read, want_aux=True)
        assert a[4] == b[3] and len(a[0]) == len(b[0]) and a[5] == b[4] == 0
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
        assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
        assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
        read += a[4]
    assert read > 3000


def test_emul_msk_many_tags_queue_refill():
    # more time_est tags in one call than the kernel's LDS queue holds (MSK_TAGQ = 36), mixed
    # with other keys: the queue is refilled in instalments and the result stays bit-exact
    rng = np.random.default_rng(11)
    nchan, lens = 19, [2600, 1400]
    total = sum(lens)
    xs = np.stack([_signal(90 + c, total, 4)[0] for c in range(nchan)])
    e = emu.MskStream(4.0, 0.04, 0.01, 1, nchan=nchan, lpw=16)  # 16 channels per wave: two waves, one ragged
    o = [orc.MskStream(4.0, 0.04, 0.01, 1) for _ in range(nchan)]
    all_tags = []
    for c in range(nchan):
        t = _rand_tags(rng, total, 150, c)
        t["key"][rng.choice(150, 30, replace=False)] = 3
        all_tags.append(t)
    k = 0
    for L in lens:
        cap = 160
        tg = np.zeros((nchan, cap), dtype=emu.TAG_DTYPE)
        cnt = np.zeros(nchan, np.int32)
        new = []
        for c in range(nchan):
            sel = all_tags[c][(all_tags[c]["offset"] >= k) & (all_tags[c]["offset"] < k + L)]
            tg[c, : len(sel)] = sel
            cnt[c] = len(sel)
            new.append(sel)
        assert (cnt > 40).any()
        r = e.step(xs[:, k:k + L], tg, cnt, want_aux=True)
        assert r["status"] == 0
        for c in range(nchan):
            ot = np.zeros(len(new[c]), dtype=orc.TAG_DTYPE)
            ot["offset"], ot["value"], ot["key"] = new[c]["offset"], new[c]["value"], new[c]["key"]
            out, o2, o3, cons = o[c].step(xs[c, k:k + L], ot, want_aux=True)
            p = r["produced"][c]
            assert p == len(out) and r["consumed"][c] == cons
            assert np.array_equal(r["syms"][c, :p].view(np.uint32), out.view(np.uint32))
            assert np.array_equal(r["mu"][c, :p].view(np.uint32), o3.view(np.uint32))
        k += L


def test_emul_msk_interp_range_is_reported():
    # a time_est tag whose value is outside the interpolator's range: upstream throws
    # std::runtime_error("mmse_fir_interpolator_cc: imu out of bounds."); here the status says so
    x, _ = _signal(5, 800)
    e = emu.MskStream(4.0, 0.04, 0.01, 1, nchan=1)
    tg = np.zeros((1, 4), dtype=emu.TAG_DTYPE)
    tg[0, 0] = (300, 5.25, 2, 0)
    r = e.step(x[None, :], tg, np.array([1], np.int32))
    assert r["status"] & 1  # MSK_ST_INTERP_RANGE
    e2 = emu.MskStream(4.0, 0.04, 0.01, 1, nchan=1)
    tg[0, 0] = (300, 0.25, 2, 0)
    assert e2.step(x[None, :], tg, np.array([1], np.int32))["status"] == 0


def _burst_like_tags(rng, total, chan, nclusters=12):
    """time_est tags the way corr_est emits them: clusters of 2-5 tags, isps (or fewer) samples
    apart, values a centre of mass in [-1, 1], now and then NaN, exactly +-1, 0, or an other key."""
    rows = []
    starts = np.sort(rng.choice(np.arange(40, total - 80), size=nclusters, replace=False))
    for s0 in starts:
        step = int(rng.choice([4, 4, 4, 1, 2, 3, 5]))
        for j in range(int(rng.integers(2, 6))):
            v = float(rng.uniform(-1, 1))
            r = rng.random()
            if r < 0.06:
                v = float("nan")
            elif r < 0.12:
                v = float(rng.choice([-1.0, 1.0, 0.0, -0.0]))
            rows.append((int(s0 + j * step), v, 2 if rng.random() > 0.05 else 1))
    rows.sort(key=lambda t: t[0])
    tags = np.zeros(len(rows), dtype=emu.TAG_DTYPE)
    tags["offset"] = [r[0] for r in rows]
    tags["value"] = [r[1] for r in rows]
    tags["key"] = [r[2] for r in rows]
    tags["chan"] = chan
    return tags


@pytest.mark.parametrize("sps,lpw", [(4.0, 8), (4.0, 4), (4.0, 16), (5.2083, 32), (5.0, 8), (4.0, 64), (4.4, 8), (3.0, 8)])
def test_emul_msk_tag_resets_inside_the_lock_step(sps, lpw):
    # osps = 1, err / mu ports not connected: the build whose lock-step runs handle time_est tags in
    # line (tag before the even iteration, tag before the odd one, clusters on consecutive pairs,
    # negative centre -> iidx - 1, NaN, +-1) -- symbols, counts and bits against the oracle
    rng = np.random.default_rng(int(sps * 100) + lpw)
    nchan, lens = 19, [2600, 37, 1800, 1, 900]
    total = sum(lens)
    xs = np.stack([_signal(150 + c, total, 4)[0] for c in range(nchan)])
    e = emu.MskStream(sps, 0.04, 0.01, 1, nchan=nchan, lpw=lpw)
    o = [orc.MskStream(sps, 0.04, 0.01, 1) for _ in range(nchan)]
    bt = [orc.BitTail() for _ in range(nchan)]
    all_tags = [_burst_like_tags(rng, total, c, nclusters=14 if c % 3 else 40) for c in range(nchan)]
    k = nsym = 0
    for L in lens:
        chunk = xs[:, k:k + L]
        cap = 256
        tg = np.zeros((nchan, cap), dtype=emu.TAG_DTYPE)
        cnt = np.zeros(nchan, np.int32)
        new = []
        for c in range(nchan):
            sel = all_tags[c][(all_tags[c]["offset"] >= k) & (all_tags[c]["offset"] < k + L)]
            tg[c, : len(sel)] = sel
            cnt[c] = len(sel)
            new.append(sel)
        # (sps 5.0: output rows of odd length -- the symbol stage then leaves in 8-byte stores;
        # sps 3.0: a tag on every other sample makes every iteration an even one, :159 -- room for that)
        r = e.step(chunk, tg, cnt, want_aux=False, out_cap={5.0: (L // 2 + 301) | 1, 3.0: L + 300}.get(sps))
        assert r["status"] == 0
        for c in range(nchan):
            ot = np.zeros(len(new[c]), dtype=orc.TAG_DTYPE)
            ot["offset"], ot["value"], ot["key"] = new[c]["offset"], new[c]["value"], new[c]["key"]
            out, _, _, cons = o[c].step(chunk[c], ot)
            p = r["produced"][c]
            assert p == len(out) and r["consumed"][c] == cons, (L, c)
            assert np.array_equal(r["syms"][c, :p].view(np.uint32), out.view(np.uint32)), (L, c)
            assert np.array_equal(r["bits"][c, :p], bt[c].process(out))
            nsym += p
        k += L
    assert nsym > nchan * total / sps * 0.9


def test_emul_bittail_sign_shortcut_on_awkward_symbols():
    # the bit tail decides by the sign of Im(sym[o] * conj(sym[o-1])) and goes through
    # fast_atan2f's table only for non-finite or underflowing operands: zeros of both signs,
    # denormals, tiny negative quadrature against a large in-phase part (-base_angle = -0 slices to
    # 1), huge values, inf and NaN -- against the oracle's full quadrature_demod -> slicer -> decoder
    import ctypes as C

    rng = np.random.default_rng(5)
    specials = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-38, -1e-38, 1e-30, -1e-30, 1e-19, -1e-19, 1e-10, -1e-10, 1.0, -1.0,
                         3.0, -3.0, 1e10, -1e10, 1e19, -1e19, 1e30, -1e30, 3e38, -3e38, np.inf, -np.inf, np.nan],
                        dtype=np.float32)
    re = rng.choice(specials, size=6000)
    im = rng.choice(specials, size=6000)
    syms = np.empty(re.size, np.complex64)
    syms.real, syms.imag = re, im  # (not re + 1j * im: 0 * inf would turn the real part into NaN)
    syms = np.concatenate([syms, (rng.normal(size=3000) + 1j * rng.normal(size=3000)).astype(np.complex64)])
    rng.shuffle(syms)
    L = emu.lib()
    L.emu_bittail_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    prev_sym = np.zeros(1, np.complex64)
    prev_bit = np.zeros(1, np.uint8)
    bt = orc.BitTail()
    k = 0
    with np.errstate(all="ignore"):
        for n in (1, 2, 5000, 1, 3996):
            bits = np.zeros(n, np.uint8)
            chunk = np.ascontiguousarray(syms[k:k + n])
            L.emu_bittail_run(chunk.ctypes.data, n, prev_sym.ctypes.data, prev_bit.ctypes.data, bits.ctypes.data)
            assert np.array_equal(bits, bt.process(chunk)), n
            k += n
