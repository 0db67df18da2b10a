"""-m gpu: the pipelined demod step (aisx_chain_*, ais_demod.work_pipelined) -- the path bench.py
times -- against the stages called one after the other on one stream (ais_demod.work), which the
other GPU tests hold to the oracle.  python/ais_demod.py:56 is the chain; same bits, symbol
counts and tags, bit for bit, is the gate."""
import numpy as np
import pytest

import oracle_py as orc
from parity import compare_bursts, compare_detections

pytestmark = pytest.mark.gpu

OPTS = dict(samples_per_symbol=4, bits_per_sec=9600.0, clockrec_gain=0.04, omega_relative_limit=0.01, fftlen=1024)


@pytest.fixture(scope="module")
def ais():
    import torch

    assert torch.cuda.is_available()
    import ais_amd

    return ais_amd


def _dev(x):
    import torch

    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def _same(ra, rb, nchan):
    pa, pb = ra["produced"].cpu().numpy(), rb["produced"].cpu().numpy()
    assert np.array_equal(pa, pb)
    ba, bb = ra["bits"].cpu().numpy(), rb["bits"].cpu().numpy()
    for c in range(nchan):
        assert np.array_equal(ba[c, :pa[c]], bb[c, :pb[c]]), c
    return int(pa.sum())


def _run_both(ais, stages, lens, xs, nchan, next_known, wrong_next_at=()):
    """serial work() on one object, work_pipelined() on its twin, over the calls `lens`; every
    AISX_CHAIN_DEPTH steps the pipelined results are collected (they rotate through that many sets)."""
    a = ais.ais_demod(OPTS, nchan=nchan, max_items=max(lens), stages=stages)
    b = ais.ais_demod(OPTS, nchan=nchan, max_items=max(lens), stages=stages)
    chunks, k = [], 0
    for L in lens:
        chunks.append(_dev(xs[:, k:k + L]))
        k += L
    nbits, pend = 0, []
    for i, x in enumerate(chunks):
        ra = a.work(x)
        ta = a.preamble_detect.tags() if ra["produced"] is not None else None
        nxt = chunks[i + 1] if (i + 1 < len(chunks) and next_known(i)) else None
        if i in wrong_next_at:  # a preparation the next call does not match: dropped, not used
            nxt = chunks[0]
        rb = b.work_pipelined(x, x_next=nxt)
        pend.append((ra, ta, rb))
        if len(pend) == 3 or i + 1 == len(chunks):
            b.synchronize()
            for j, (ra, ta, rb) in enumerate(pend):
                if ra["produced"] is None:  # less than one fftlen-vector so far: nothing reaches corr_est
                    assert int(rb["produced"].abs().sum()) == 0
                    assert len(b.step_tags(rb["step"])) == 0
                    continue
                nbits += _same(ra, rb, nchan)
                # (tags of a step whose front end emitted nothing are not in the rotation)
                back = sum(1 for q in pend[j + 1:] if q[0]["produced"] is not None)
                assert ta.tobytes() == b.preamble_detect.tags(back=back).tobytes(), (i, j)
                assert ta.tobytes() == b.step_tags(rb["step"]).tobytes(), (i, j)  # ... by step number: no counting
            pend = []
    assert b.clockrec.last_status() == 0
    return nbits


def test_pipelined_step_equals_the_serial_chain_ragged(ais):
    # ragged calls: lengths that are not whole vectors (estimates prepared BEHIND the pass), a first
    # call shorter than one vector (nothing reaches the correlator), whole vectors in a row
    # (estimates two calls ahead), next input known for some steps only
    import synth

    nchan = 70
    lens = [600, 4096, 2048, 1000, 24, 5000, 3 * 1024 + 7, 8192, 8192, 1024, 2041]
    xs = np.stack([synth.make_channel(4100 + c, sum(lens), "S", 4, amp=0.3, cfo_max=500.0)[0] for c in range(nchan)])
    nbits = _run_both(ais, "stock", lens, xs, nchan, lambda i: i % 4 != 2, wrong_next_at=(4, 7))
    assert nbits > nchan * (sum(lens) - 2048) / 4 * 0.9


def test_pipelined_core_chain(ais):
    # corr_est -> msk only (fs = agc = NULL): the chain BASELINE.json's metric names
    import synth

    nchan = 33
    lens = [8192, 5000, 12288, 100, 4096]
    xs = np.stack([synth.make_channel(4300 + c, sum(lens), "S", 4, amp=1.0, cfo_max=3.0)[0] for c in range(nchan)])
    nbits = _run_both(ais, "core", lens, xs, nchan, lambda i: True)
    assert nbits > nchan * sum(lens) / 4 * 0.9


def test_two_alternating_input_buffers_against_the_oracle(ais):
    # A live source runs one block ahead through two buffers A, B, A, B ...: step k is issued when
    # block k + 1 has arrived in the other buffer (x_next), and a buffer is refilled only after
    # wait_input() of the step that read it.  The prepared estimates must belong to the block they
    # were prepared for even though the POINTER repeats every other step: checked
    # against the oracle's chain, which knows nothing of buffers.
    import torch
    import synth
    from ais_amd import _lib
    import ctypes as C

    nchan, K, T, steps = 24, 6, 16384, 6
    made = [synth.make_channel(4500 + c, T * steps, "S", 4, amp=0.3, cfo_max=500.0) for c in range(nchan)]
    xs = np.stack([m[0] for m in made])
    dem = ais.ais_demod(OPTS, nchan=nchan, max_items=T, stages="stock")
    tmpl = np.asarray(dem.mod_vector, dtype=np.complex64)
    thr = dem.preamble_detect.threshold()
    ora = [orc.Demod(4, tmpl, stages=3) for _ in range(K)]
    gb, ob_all = [[] for _ in range(K)], [[] for _ in range(K)]
    lone = near_thr = 0
    buf = [torch.empty((nchan, T), dtype=torch.complex64, device="cuda") for _ in range(2)]
    buf[0].copy_(_dev(xs[:, :T]))
    ndet = 0
    for s in range(steps):
        if s + 1 < steps:
            # refill the other buffer: its last reader was step s - 1
            if s >= 1:
                _lib.check(_lib.lib().aisx_chain_wait_input(dem._chain_handle(), s - 1, C.c_void_p(torch.cuda.current_stream().cuda_stream), 0),
                           "wait_input")
            buf[(s + 1) & 1].copy_(_dev(xs[:, (s + 1) * T:(s + 2) * T]))
        r = dem.work_pipelined(buf[s & 1], x_next=buf[(s + 1) & 1] if s + 1 < steps else None)
        dem.wait(host=True)
        tags = dem.preamble_detect.tags()
        prod, bits = r["produced"].cpu().numpy(), r["bits"].cpu().numpy()
        for c in range(K):
            ob, _, ot = ora[c].step(xs[c, s * T:(s + 1) * T])
            d = compare_detections(tags[tags["chan"] == c], ot, thr)
            ndet += d["matched"]
            lone += d["lone"]
            near_thr += d["lone_near_threshold"]
            assert d["mag_rel_max"] <= 1e-5 and d["time_est_abs_max"] <= 1e-4, (s, c, d)
            gb[c].append(bits[c, :prod[c]].copy())
            ob_all[c].append(ob)
    assert lone == near_thr and lone <= 2, (lone, near_thr)  # (stale estimates would move every detection)
    ncmp = same = 0
    for c in range(K):
        a, b, _ = compare_bursts(np.concatenate(gb[c]), np.concatenate(ob_all[c]), made[c][1])
        ncmp, same = ncmp + a, same + b
    print("two alternating buffers: %d detections matched, %d of %d bursts identical in place" % (ndet, same, ncmp))
    assert ndet > 5 * K * steps and ncmp > 2 * K * steps and same >= ncmp - 2 * max(1, lone) and dem.clockrec.last_status() == 0


def test_chain_argument_checks(ais):
    import synth

    nchan, T = 4, 4096
    dem = ais.ais_demod(OPTS, nchan=nchan, max_items=T, stages="stock")
    x = _dev(np.stack([synth.make_channel(4700 + c, T, "S", 4, amp=0.3)[0] for c in range(nchan)]))
    with pytest.raises(ValueError):
        dem.work_pipelined(_dev(np.zeros((nchan, 2 * T), np.complex64)))  # n > max_items
    with pytest.raises(ValueError):
        dem.wait(step=5)  # never issued
    r = dem.work_pipelined(x)
    dem.wait(host=True)
    assert r["step"] == 0 and int(r["produced"].min()) > (T - 1024) // 4 - 64
    with pytest.raises(ValueError):
        dem.corr_output(3)
    assert dem.corr_output(0, 1, 2).shape == (2, T)



def test_chain_create_checks_the_borrowed_handles(ais):
    """aisx_chain_create: a stage handle built for another channel count or a smaller capacity is refused
    (AISX_ERR_INVALID -> ValueError) instead of writing past the chain's rows at the first step."""
    import ctypes as C
    from ais_amd import _lib

    L = _lib.lib()
    tmpl = np.ones(112, np.complex64)

    def handles(nchan_corr=8, mi_msk=4096 + 1024, agc_w=512, fft_fs=1024):
        fs, agc, corr, msk = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        assert L.aisx_freqsync_create(C.byref(fs), 38400.0, 9600.0, fft_fs, 8, 4096) == 0
        assert L.aisx_agc_create(C.byref(agc), agc_w, 2.0, 8, 4096 + 1024) == 0
        assert L.aisx_corr_create(C.byref(corr), tmpl.ctypes.data_as(C.c_void_p), 112, 4.0, 1, 0.9, nchan_corr, 4096 + 1024, 512) == 0
        assert L.aisx_msk_create(C.byref(msk), 4.0, 0.04, 0.01, 1, 8, mi_msk) == 0
        return fs, agc, corr, msk

    def create(hs):
        ch = C.c_void_p()
        rc = L.aisx_chain_create(C.byref(ch), hs[0], hs[1], hs[2], hs[3], 8, 4096, 1024)
        msg = L.aisx_last_error().decode()
        if rc == 0:
            L.aisx_chain_destroy(ch)
        L.aisx_freqsync_destroy(hs[0]), L.aisx_agc_destroy(hs[1]), L.aisx_corr_destroy(hs[2]), L.aisx_msk_destroy(hs[3])
        return rc, msg

    assert create(handles())[0] == 0
    rc, msg = create(handles(nchan_corr=16))
    assert rc == _lib.AISX_ERR_INVALID and "corr_est" in msg and "nchan" in msg
    rc, msg = create(handles(mi_msk=4096))
    assert rc == _lib.AISX_ERR_INVALID and "msk_timing_recovery" in msg and "max_items" in msg
    rc, msg = create(handles(agc_w=500))
    assert rc == _lib.AISX_ERR_INVALID and "agc" in msg and "window" in msg


def test_wait_prepass_is_an_event_wait_unless_a_head_start_is_asked_for(ais):
    blk = ais.msk_timing_recovery_cc(4.0, 0.04, 0.01, 1, nchan=4, max_items=4096)
    from ais_amd import _lib

    L = _lib.lib()
    assert L.aisx_msk_set_head_start(blk._h, 20) == 0 and L.aisx_msk_set_head_start(blk._h, 0) == 0
    assert L.aisx_msk_set_head_start(blk._h, -1) == _lib.AISX_ERR_INVALID


def test_chain_sets_its_placement_claims_and_gives_the_handles_theirs_back(ais):
    """aisx_chain_create derives LDS claims for the front-end kernel and the phase walk from the part and the recovery's launch
    (aisx_chain.hip: chain_front_claim) and sets them on the borrowed handles; what the caller had set before comes back when
    the chain is destroyed.  Few channels = the recovery leaves more than half of the CUs free = both claims non-zero."""
    import ctypes as C
    from ais_amd import _lib

    L = _lib.lib()
    tmpl = np.ones(112, np.complex64)
    fs, agc, corr, msk, ch = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert L.aisx_freqsync_create(C.byref(fs), 38400.0, 9600.0, 1024, 8, 4096) == 0
    assert L.aisx_agc_create(C.byref(agc), 512, 2.0, 8, 4096 + 1024) == 0
    assert L.aisx_corr_create(C.byref(corr), tmpl.ctypes.data_as(C.c_void_p), 112, 4.0, 1, 0.9, 8, 4096 + 1024, 512) == 0
    assert L.aisx_msk_create(C.byref(msk), 4.0, 0.04, 0.01, 1, 8, 4096 + 1024) == 0
    assert L.aisx_agc_set_lds_claim(agc, 5 * 1024) == 0 and L.aisx_freqsync_set_walk_lds_claim(fs, 7 * 1024) == 0

    def claims():
        a, w, ua, uw = C.c_int(-1), C.c_int(-1), C.c_int(-1), C.c_int(-1)
        assert L.aisx_agc_get_lds_claim(agc, C.byref(a), C.byref(ua)) == 0
        assert L.aisx_freqsync_get_walk_lds_claim(fs, C.byref(w), C.byref(uw)) == 0
        assert ua.value > 0 and uw.value > 0
        return a.value, w.value

    wgs, lds = C.c_int(0), C.c_int(0)
    assert L.aisx_msk_placement(msk, C.byref(wgs), C.byref(lds)) == 0 and wgs.value == 1 and 0 < lds.value < 160 * 1024
    assert claims() == (5 * 1024, 7 * 1024)
    assert L.aisx_chain_create(C.byref(ch), fs, agc, corr, msk, 8, 4096, 1024) == 0
    a, w = claims()
    assert a > 160 * 1024 - lds.value - 16 * 1024 and w > a  # (none of either fits beside a recovery workgroup)
    assert L.aisx_chain_destroy(ch) == 0
    assert claims() == (5 * 1024, 7 * 1024)
    assert L.aisx_freqsync_set_walk_lds_claim(fs, -1) == _lib.AISX_ERR_INVALID
    for h, d in ((fs, L.aisx_freqsync_destroy), (agc, L.aisx_agc_destroy), (corr, L.aisx_corr_destroy), (msk, L.aisx_msk_destroy)):
        d(h)
