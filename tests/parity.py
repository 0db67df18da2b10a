"""Shared parity helpers for the test-suite (tolerances of BASELINE.json's
north_star: tag positions within +-1 sample, peak magnitude within 1e-5
relative, time_est within 1e-4 absolute)."""
import numpy as np

MAG_RTOL = 1e-5
TIME_ATOL = 1e-4
PHASE_ATOL = 2e-4


def tag_groups(tags, port1_flag=0x100):
    """Split a tag array (fields offset,value,key) into detection groups
    [(corr_start_off, mag, phase, center, mark_off)] for port-0 tags."""
    groups = []
    cur = None
    for t in tags:
        key = int(t["key"])
        if "port" in t.dtype.names and int(t["port"]) != 0:
            continue
        if key & port1_flag:
            continue
        if key == 0:
            cur = dict(start=int(t["offset"]), mag=float(t["value"]))
            groups.append(cur)
        elif cur is not None:
            if key == 1:
                cur["phase"] = float(t["value"])
                cur["mark"] = int(t["offset"])
            elif key == 2:
                cur["center"] = float(t["value"])
            elif key == 3:
                cur["est"] = float(t["value"])
    return groups


def assert_tags_match(got, want, exact_offsets=True):
    g, w = tag_groups(got), tag_groups(want)
    assert len(g) == len(w), "detections: got %d want %d\n%s\n%s" % (len(g), len(w), g[:8], w[:8])
    for a, b in zip(g, w):
        if exact_offsets:
            assert a["start"] == b["start"], (a, b)
        else:
            assert abs(a["start"] - b["start"]) <= 1, (a, b)
        assert a["mark"] - a["start"] == b["mark"] - b["start"]
        assert abs(a["mag"] - b["mag"]) <= MAG_RTOL * abs(b["mag"]), (a, b)
        assert abs(a["est"] - b["est"]) <= MAG_RTOL * abs(b["est"]), (a, b)
        if a["start"] == b["start"]:
            assert abs(a["center"] - b["center"]) <= TIME_ATOL, (a, b)
            d = abs(a["phase"] - b["phase"])
            assert min(d, abs(d - 2 * np.pi)) <= PHASE_ATOL, (a, b)
    return len(g)


def unit_template(rng, N):
    return np.exp(1j * rng.uniform(-np.pi, np.pi, N)).astype(np.complex64)


def planted(rng, nchan, n, tmpl, positions, noise=0.05, amp=1.0):
    x = (noise * (rng.normal(size=(nchan, n)) + 1j * rng.normal(size=(nchan, n)))).astype(np.complex64)
    N = tmpl.size
    for c, plist in enumerate(positions):
        for p in plist:
            lo, hi = max(p, 0), min(p + N, n)
            if hi > lo:
                x[c, lo:hi] += (amp * tmpl[lo - p:hi - p] * np.exp(1j * rng.uniform(-3, 3))).astype(np.complex64)
    return x


def assert_decoded_bursts_identical(got_bits, want_bits, infos, min_frac_equal=0.97):
    """Chain-level gate: every burst the oracle's bit stream contains
    must appear, bit for bit, at the same position in the GPU's stream.  Bits
    demodulated from noise between bursts are not compared one by one: a
    time_est that differs in its last place (FFT rounding) legitimately flips a
    few of them; only their overall agreement is bounded.  Returns
    (bursts compared, bursts transmitted)."""
    import synth

    got_bits = np.asarray(got_bits, dtype=np.uint8)
    want_bits = np.asarray(want_bits, dtype=np.uint8)
    assert got_bits.size == want_bits.size
    ncmp = 0
    for inf in infos:
        pat = np.asarray(inf["data_bits"], dtype=np.uint8)
        for pos in synth.find_bits(want_bits, pat):
            assert np.array_equal(got_bits[pos:pos + pat.size], pat), "burst at bit %d differs" % pos
            ncmp += 1
    if want_bits.size:
        # achieved (round 2, 112 channel runs of ~12 300 bits): 108 channels 1.0, worst 0.973 -- one
        # symbol slip in the noise before a burst shifts everything up to the next tag reset
        frac = float(np.mean(got_bits == want_bits))
        FRACTIONS.append(frac)
        assert frac >= min_frac_equal, "only %.4f of the bits agree" % frac
    return ncmp, len(infos)


FRACTIONS = []  # per-channel agreement of the last calls (callers assert on the aggregate)


def assert_aggregate_agreement(min_mean=0.9985, min_exact_share=0.7):
    """Over the channels compared since the last call: mean bit agreement and the share of
    channels whose whole bit stream is identical (achieved per 24-channel test: means 1.0 ...
    0.99887, identical streams 24 ... 18 of 24 -- which channels slip a symbol in the noise
    depends on the last place of a time_est and moves with any change of rounding upstream)."""
    f = np.array(FRACTIONS)
    del FRACTIONS[:]
    assert f.size and f.mean() >= min_mean and np.mean(f == 1.0) >= min_exact_share, (f.mean(), np.mean(f == 1.0), f.min())


def compare_detections(got, want, thr=None, near_rel=2e-5):
    """Non-asserting comparison of two tag lists of one channel and one call (the gates of
    BASELINE.md section 3, as numbers): detection groups are paired when their corr_start offsets
    are within +-1 sample; a group present on one side only is `lone` (and `lone_near_threshold`
    if its peak is within 2e-5 relative of the correlator's threshold: both correlators agree to
    ~3e-7, so only such a peak can be seen by one and not by the other)."""
    g, w = tag_groups(got), tag_groups(want)
    r = dict(detections=len(w), matched=0, offsets_equal=0, lone=0, lone_near_threshold=0, mag_rel_max=0.0,
             time_est_abs_max=0.0, phase_abs_max=0.0)
    i = j = 0
    while i < len(g) or j < len(w):
        a = g[i] if i < len(g) else None
        b = w[j] if j < len(w) else None
        if a is not None and b is not None and abs(a["start"] - b["start"]) <= 1:
            r["matched"] += 1
            r["mag_rel_max"] = max(r["mag_rel_max"], abs(a["mag"] - b["mag"]) / max(abs(b["mag"]), 1e-30),
                                   abs(a["est"] - b["est"]) / max(abs(b["est"]), 1e-30))
            if a["start"] == b["start"]:
                r["offsets_equal"] += 1
                r["time_est_abs_max"] = max(r["time_est_abs_max"], abs(a["center"] - b["center"]))
                d = abs(a["phase"] - b["phase"])
                r["phase_abs_max"] = max(r["phase_abs_max"], min(d, abs(d - 2 * np.pi)))
            i += 1
            j += 1
            continue
        lone = a if (b is None or (a is not None and a["start"] < b["start"])) else b
        r["lone"] += 1
        if thr is not None and abs(lone["mag"] - thr) <= near_rel * thr:
            r["lone_near_threshold"] += 1
        if lone is a:
            i += 1
        else:
            j += 1
    return r


def compare_bursts(got_bits, want_bits, infos, slack=4):
    """Every burst the reference side's bit stream holds (its HDLC-framed data bits found in
    `want_bits`) is looked up in `got_bits`: bit for bit at the same position, or within +-slack
    positions (a detection seen by one chain only re-times the loop and can move the symbol count
    by one in the noise before the burst).  Returns (compared, identical_in_place, identical_within_slack)."""
    import synth

    got_bits = np.asarray(got_bits, dtype=np.uint8)
    want_bits = np.asarray(want_bits, dtype=np.uint8)
    ncmp = same = near = 0
    for inf in infos:
        pat = np.asarray(inf["data_bits"], dtype=np.uint8)
        for pos in synth.find_bits(want_bits, pat):
            ncmp += 1
            if np.array_equal(got_bits[pos:pos + pat.size], pat):
                same += 1
                near += 1
            elif any(np.array_equal(got_bits[max(pos + d, 0):max(pos + d, 0) + pat.size], pat) for d in range(-slack, slack + 1)):
                near += 1
    return ncmp, same, near


def freqest_cases(rng, fftlen, sample_rate, data_rate, nchan=3, nvec=6):
    """Spectra for freqest::work at any vector length: noise vectors with a planted peak pair `offset` bins apart, all-zero
    vectors (the reference keeps the previous vector's maxpos: lib/freqest_impl.cc:68 vs :74), a tie between two pairs (the
    first strict maximum wins), a peak in the last searched bin."""
    offset = int(fftlen * (np.float32(data_rate) / np.float32(sample_rate)))
    span = fftlen - offset
    v = (rng.normal(size=(nchan, nvec, fftlen)) + 1j * rng.normal(size=(nchan, nvec, fftlen))).astype(np.complex64) * 0.05
    for c in range(nchan):
        for k in range(nvec):
            if span > 0 and k % 3 != 2:
                j = int(rng.integers(0, span))
                v[c, k, j] += 4 + c
                v[c, k, j + offset] += 3j
    v[0, 1] = 0  # stale maxpos behind vector 0
    v[1, 0] = 0  # a call that opens with silence: maxpos 0
    if span > 2:
        v[2, 3] = 0
        v[2, 3, 1] = v[2, 3, 1 + offset] = 1  # a tie: bins 1 and span - 1
        v[2, 3, span - 1] = v[2, 3, span - 1 + offset] = 1
        v[2, 4] = 0
        v[2, 4, span - 1] = 2
    return v
