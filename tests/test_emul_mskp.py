"""The time-parallel timing recovery (gr-ais_amd/csrc/k_mskp.h: prepass, units, join, gather; and the
serial kernel of k_msk.h as the join, MskParams::ff) under the CPU lane model vs the oracle: symbols,
bits and item counts must be BIT-identical whatever the units made of their restart points -- a unit
whose junction does not check is thrown away and its stretch is run serially."""
import numpy as np
import pytest

import emul_py as emu
import oracle_py as orc
import synth


def _tags_with_pairs(rng, total, chan, sps, pair_every, neg_frac, extra_random, nan_at=None):
    """time_est tags as corr_est leaves them on bursts: pairs one symbol apart; neg_frac of the pairs get a
    small negative centre (mu close to 1: the iteration behind tag A may advance by three items and step
    over tag B, which then blocks every later tag of the call, reference :140-142)."""
    offs, vals = [], []
    pos = int(rng.integers(20, pair_every))
    while pos < total - 20:
        v = rng.uniform(-0.5, 0.5)
        if rng.random() < neg_frac:
            v = -abs(v) * 0.2
        offs += [pos, pos + int(round(sps))]
        vals += [v, v + rng.normal(0, 0.02)]
        pos += int(rng.integers(pair_every // 2, pair_every * 3 // 2))
    for _ in range(extra_random):
        offs.append(int(rng.integers(10, total - 10)))
        vals.append(rng.uniform(-0.9, 0.9))
    t = np.zeros(len(offs), dtype=emu.TAG_DTYPE)
    t["offset"], t["value"], t["key"], t["chan"] = offs, vals, 2, chan
    t = t[np.argsort(t["offset"], kind="stable")]
    if nan_at is not None and len(t) > nan_at:
        t["value"][nan_at] = np.nan
    return t


def _run(nchan, lens, seed, join, Q=0, sps=4.0, smax=16, min_gap=128, pair_every=400, neg_frac=0.3, extra_random=6,
         nan_at=None, first_tag_at_zero=False, lpw=8):
    rng = np.random.default_rng(seed)
    total = sum(lens)
    xs = np.stack([synth.make_channel(50 + c + seed, total, "P", 4, amp=1.0, cfo_max=50.0)[0] for c in range(nchan)])
    e = emu.MskStream(sps, 0.04, 0.01, 1, nchan=nchan, lpw=lpw, tp_smax=smax, tp_min_gap=min_gap, max_noutput=Q, tp_join=join)
    o = [orc.MskStream(sps, 0.04, 0.01, 1, max_noutput=Q) for _ in range(nchan)]
    bt = [orc.BitTail() for _ in range(nchan)]
    all_tags = [_tags_with_pairs(rng, total, c, sps, pair_every, neg_frac, extra_random, nan_at) for c in range(nchan)]
    if first_tag_at_zero:  # a tag at nitems_read with a negative centre: the loop reads in[-1] (:150-153)
        for c in range(nchan):
            all_tags[c]["offset"][0] = lens[0]  # = nitems_read of the second call, give or take the carried items
            all_tags[c]["value"][0] = -0.4
            all_tags[c] = all_tags[c][np.argsort(all_tags[c]["offset"], kind="stable")]
    k = 0
    for L in lens:
        chunk = xs[:, k:k + L]
        cap = max(len(t) for t in all_tags) + 1
        tg = np.zeros((nchan, cap), dtype=emu.TAG_DTYPE)
        cnt = np.zeros(nchan, np.int32)
        new = []
        for c in range(nchan):
            sel = all_tags[c][(all_tags[c]["offset"] >= k) & (all_tags[c]["offset"] < k + L)]
            tg[c, : len(sel)] = sel
            cnt[c] = len(sel)
            new.append(sel)
        r = e.step(chunk, tg, cnt)
        assert r["status"] == 0
        for c in range(nchan):
            ot = np.zeros(len(new[c]), dtype=orc.TAG_DTYPE)
            ot["offset"], ot["value"], ot["key"] = new[c]["offset"], new[c]["value"], new[c]["key"]
            out, _, _, cons = o[c].step(chunk[c], ot)
            p = r["produced"][c]
            assert p == len(out) and r["consumed"][c] == cons, (c, L, p, len(out), r["consumed"][c], cons)
            assert np.array_equal(r["syms"][c, :p].view(np.uint32), out.view(np.uint32)), (c, L)
            assert np.array_equal(r["bits"][c, :p], bt[c].process(out)), (c, L)
        k += L
    return e.tp_stats()


@pytest.mark.parametrize("join", [0, 1])
@pytest.mark.parametrize("Q", [0, 100])
def test_units_and_join_bit_exact(join, Q):
    st = _run(3, [6000, 37, 3000, 1, 2500], seed=1, join=join, Q=Q)
    # the restart points were used: most symbols came from units, and what they produced is the oracle's
    assert st["restart_points"] >= 20 and st["units_accepted"] >= 0.6 * st["restart_points"]
    assert st["symbols_from_units"] >= 0.4 * st["symbols"]


@pytest.mark.parametrize("join", [0, 1])
def test_junctions_that_do_not_check_are_run_serially(join):
    """Nine pairs in ten with a small negative centre (tag B is often stepped over and blocks the rest of the
    call), a NaN tag, a tag with a negative centre right at nitems_read of the second call: units are thrown
    away, the result is still the oracle's bit for bit."""
    st = _run(4, [8000, 4000], seed=2, join=join, Q=100, neg_frac=0.9, pair_every=300, nan_at=11, first_tag_at_zero=True)
    assert st["restart_points"] >= 40
    assert st["units_accepted"] < st["restart_points"]  # some junctions did fail ...
    assert st["units_accepted"] >= 0.3 * st["restart_points"]  # ... and the others were taken over


@pytest.mark.parametrize("join", [0, 1])
def test_unblocked_contract_a_stale_tag_blocks_to_the_end_of_the_step(join):
    # without max_noutput_items a stale tag blocks every later tag of the step: the join runs that tail serially
    st = _run(3, [8000, 4000], seed=3, join=join, Q=0, neg_frac=0.9, pair_every=300)
    assert st["restart_points"] > 40


@pytest.mark.parametrize("sps", [5.2083, 3.0])
def test_other_samples_per_symbol(sps):
    # (sps < 4 with max_noutput_items: no units -- a call boundary could hand a tag out twice -- the join alone)
    st = _run(2, [5000, 3000], seed=4, join=1, Q=64, sps=sps)
    if sps < 4:
        assert st["restart_points"] == 0
    else:
        assert st["units_accepted"] > 0


@pytest.mark.parametrize("lpw", [8, 64])
def test_max_noutput_items_in_the_serial_kernel(lpw):
    # gr::block::set_max_noutput_items(): the serial kernel (no restart points at all) under the same contract
    _run(3, [6000, 37, 3000], seed=5, join=0, Q=256, smax=-1, lpw=lpw)
