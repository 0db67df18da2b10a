"""-m gpu: the time-parallel timing recovery (gr-ais_amd/csrc/k_mskp.h; aisx_msk_set_time_parallel) and
gr::block::set_max_noutput_items() (aisx_msk_set_max_noutput_items) on the device, through the C ABI:
symbols, bits and counts bit-identical to the oracle (reference lib/msk_timing_recovery_cc_impl.cc:107-206
under the stream contract), with junctions that check and junctions that are made to fail; and the
pipelined chain with the mode on equal, bit for bit, to the chain with the serial kernel."""
import numpy as np
import pytest

import oracle_py as orc
from test_emul_mskp import _tags_with_pairs

pytestmark = pytest.mark.gpu

OPTS = dict(samples_per_symbol=4, bits_per_sec=9600.0, clockrec_gain=0.04, omega_relative_limit=0.01, fftlen=1024)


@pytest.fixture(scope="module")
def ais():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a visible MI355X"
    import ais_amd

    return ais_amd


def _dev(x):
    import torch

    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def _stream(ais, nchan, lens, seed, join, Q, neg_frac, pair_every, nan_at=None, every=1, sps=4.0):
    import torch
    import synth

    rng = np.random.default_rng(seed)
    total = sum(lens)
    xs = np.stack([synth.make_channel(900 + c + seed, total, "P", 4, amp=1.0, cfo_max=50.0)[0] for c in range(nchan)])
    blk = ais.msk_timing_recovery_cc(sps, 0.04, 0.01, 1, nchan=nchan, max_items=max(lens))
    blk.set_time_parallel(64, join_kernel=join, max_unit_items=16384)
    blk.set_max_noutput_items(Q)
    o = {c: orc.MskStream(sps, 0.04, 0.01, 1, max_noutput=Q) for c in range(0, nchan, every)}
    bt = {c: orc.BitTail() for c in o}
    all_tags = [_tags_with_pairs(rng, total, c, sps, pair_every, neg_frac, 6, nan_at) for c in range(nchan)]
    for c in range(nchan):
        all_tags[c].dtype.names  # (TAG_DTYPE of the lane model = ais.TAG_DTYPE's layout)
    k, stats = 0, []
    for L in lens:
        cap = max(len(t) for t in all_tags) + 1
        tg = np.zeros((nchan, cap), dtype=ais.TAG_DTYPE)
        cnt = np.zeros(nchan, np.int32)
        new = []
        for c in range(nchan):
            sel = all_tags[c][(all_tags[c]["offset"] >= k) & (all_tags[c]["offset"] < k + L)]
            for f in ("offset", "value", "key", "chan"):
                tg[f][c, : len(sel)] = sel[f]
            cnt[c] = len(sel)
            new.append(sel)
        d_tags = torch.as_tensor(tg.view(np.uint8).reshape(nchan, -1).copy()).cuda()
        d_cnt = torch.as_tensor(cnt).cuda()
        r = blk.work(_dev(xs[:, k:k + L]), tags_ptrs=(d_tags.data_ptr(), d_cnt.data_ptr(), cap))
        assert blk.last_status() == 0
        stats.append(blk.restart_stats())
        prod = r["produced"].cpu().numpy()
        syms, bits = r["syms"].cpu().numpy(), r["bits"].cpu().numpy()
        for c in o:
            ot = np.zeros(len(new[c]), dtype=orc.TAG_DTYPE)
            ot["offset"], ot["value"], ot["key"] = new[c]["offset"], new[c]["value"], new[c]["key"]
            out, _, _, _ = o[c].step(xs[c, k:k + L], ot)
            p = prod[c]
            assert p == len(out), (c, L)
            assert np.array_equal(syms[c, :p].view(np.uint32), out.view(np.uint32)), (c, L)
            assert np.array_equal(bits[c, :p], bt[c].process(out)), (c, L)
        k += L
    return stats


@pytest.mark.parametrize("join", [1, 0])
@pytest.mark.parametrize("Q", [0, 256])
def test_time_parallel_stream_bit_exact(ais, join, Q):
    stats = _stream(ais, 70, [30000, 37, 12000, 1, 9000], seed=1, join=join, Q=Q, neg_frac=0.3, pair_every=900)
    s = stats[0]
    assert s["calls"] >= 1 and s["restart_points"] >= 70 * 20
    assert s["units_taken"] >= 0.6 * s["restart_points"]  # the mode was at work ...
    assert s["links_equal"] >= 0.8 * max(1, s["links"])  # ... and units end where the next ones assumed


@pytest.mark.parametrize("join", [1, 0])
def test_time_parallel_with_failing_junctions(ais, join):
    """Nine pairs in ten with a small negative centre (tag B is stepped over now and then and blocks the later
    tags of the call, reference :140-142), a NaN tag: units are thrown away, the results stay the oracle's."""
    stats = _stream(ais, 66, [40000, 20000], seed=2, join=join, Q=128, neg_frac=0.9, pair_every=500, nan_at=17)
    s = stats[0]
    assert s["restart_points"] >= 66 * 40
    assert s["units_taken"] < s["restart_points"]
    assert s["units_taken"] >= 0.3 * s["restart_points"]


@pytest.mark.parametrize("Q", [256, 2048])
def test_max_noutput_items_serial_kernel(ais, Q):
    # set_max_noutput_items() with the serial kernel alone (restart points off)
    import torch
    import synth

    nchan, lens = 40, [20000, 37, 9000]
    total = sum(lens)
    xs = np.stack([synth.make_channel(300 + c, total, "P", 4, amp=1.0, cfo_max=50.0)[0] for c in range(nchan)])
    blk = ais.msk_timing_recovery_cc(4.0, 0.04, 0.01, 1, nchan=nchan, max_items=max(lens))
    blk.set_max_noutput_items(Q)
    assert blk.max_noutput_items() == Q
    o = [orc.MskStream(4.0, 0.04, 0.01, 1, max_noutput=Q) for _ in range(nchan)]
    k = 0
    for L in lens:
        r = blk.work(_dev(xs[:, k:k + L]), want_aux=True)
        assert blk.last_status() == 0
        prod, syms, mu = r["produced"].cpu().numpy(), r["syms"].cpu().numpy(), r["mu"].cpu().numpy()
        for c in range(nchan):
            out, _, o3, _ = o[c].step(xs[c, k:k + L], np.zeros(0, orc.TAG_DTYPE), want_aux=True)
            assert prod[c] == len(out)
            assert np.array_equal(syms[c, :prod[c]].view(np.uint32), out.view(np.uint32))
            assert np.array_equal(mu[c, :prod[c]].view(np.uint32), o3.view(np.uint32))
        k += L


@pytest.mark.parametrize("join", [1, 0])
def test_pipelined_chain_time_parallel_equals_the_serial_kernel(ais, join):
    """python/ais_demod.py:56 through aisx_chain_step: one object with the serial kernel, its twin with the
    time-parallel recovery (units on their own stream, beside the previous step's join): the same bits,
    symbol counts and tags, bit for bit, over three full-length steps; and 16 channels against the oracle."""
    import bench

    nchan, T, steps = 96, 65536, 3
    tmpl = bench.make_template("S", 4)
    import torch

    x = bench.make_input(nchan, T, "S", 4, torch.device("cuda", 0), 0, True)
    Q = 256
    a = ais.ais_demod(OPTS, nchan=nchan, max_items=T, stages="stock", preamble_symbols=tmpl, fused_front_end=True)
    b = ais.ais_demod(OPTS, nchan=nchan, max_items=T, stages="stock", preamble_symbols=tmpl, fused_front_end=True)
    a.clockrec.set_max_noutput_items(Q)
    b.clockrec.set_max_noutput_items(Q)
    b.clockrec.set_time_parallel(64, join_kernel=join, max_unit_items=16384)
    ra, rb = [], []
    for _ in range(steps):
        ra.append(a.work_pipelined(x, x_next=x))
        rb.append(b.work_pipelined(x, x_next=x))
    a.synchronize()
    b.synchronize()
    st = b.clockrec.restart_stats()
    assert st["calls"] == steps and st["units_taken"] >= 0.6 * st["restart_points"] > 0
    nbits = 0
    for qa, qb in zip(ra, rb):
        pa, pb = qa["produced"].cpu().numpy(), qb["produced"].cpu().numpy()
        assert np.array_equal(pa, pb)
        ba, bb = qa["bits"].cpu().numpy(), qb["bits"].cpu().numpy()
        for c in range(nchan):
            assert np.array_equal(ba[c, :pa[c]], bb[c, :pb[c]]), c
        nbits += int(pa.sum())
    assert nbits > 0.9 * nchan * steps * T / 4
    assert a.preamble_detect.tags().tobytes() == b.preamble_detect.tags().tobytes()
    # the oracle's chain under the same scheduling, 16 channels, the last step's bits
    xh = x[:16].cpu().numpy()
    gb = rb[-1]["bits"][:16].cpu().numpy()
    gp = rb[-1]["produced"][:16].cpu().numpy()
    same = 0
    for c in range(16):
        dem = orc.Demod(4, tmpl, stages=3, max_noutput=Q)
        for _ in range(steps):
            bits, _, _ = dem.step(xh[c])
        same += int(len(bits) == gp[c] and np.array_equal(bits, gb[c, : gp[c]]))
    assert same >= 12  # (a time_est that differs in its last place may slip a symbol in the noise: parity.py)


def test_time_parallel_at_the_benchmark_size_equals_the_serial_kernel(ais):
    """BASELINE config 3's shape (4096 channels x 65536 samples, the stock template) through the pipelined chain with
    the time-parallel recovery on (Q = 256, units <= 16384 items) against its twin with the serial kernel under the
    same Q: symbol counts and bits of ALL 4096 channels and the tags, over two steps -- the size-independent
    property that the restart points change nothing; and most symbols must have come from units."""
    import torch

    import bench

    nchan, T, steps, Q = 4096, 65536, 2, 256
    tmpl = bench.make_template("S", 4)
    x = bench.make_input(nchan, T, "S", 4, torch.device("cuda", 0), 0, True)
    a = ais.ais_demod(OPTS, nchan=nchan, max_items=T, stages="stock", preamble_symbols=tmpl, fused_front_end=True)
    b = ais.ais_demod(OPTS, nchan=nchan, max_items=T, stages="stock", preamble_symbols=tmpl, fused_front_end=True)
    a.clockrec.set_max_noutput_items(Q)
    b.clockrec.set_max_noutput_items(Q)
    b.clockrec.set_time_parallel(64, join_kernel=1, max_unit_items=16384)
    ra = [a.work_pipelined(x, x_next=x) for _ in range(steps)]
    rb = [b.work_pipelined(x, x_next=x) for _ in range(steps)]
    a.synchronize()
    b.synchronize()
    assert a.clockrec.last_status() == 0 and b.clockrec.last_status() == 0
    st = b.clockrec.restart_stats()
    assert st["calls"] == steps and st["restart_points"] > 30 * nchan
    assert st["units_taken"] >= 0.9 * st["restart_points"] and st["symbols_from_units"] >= 0.8 * nchan * T / 4
    for qa, qb in zip(ra, rb):
        assert torch.equal(qa["produced"], qb["produced"])
        w = int(qa["produced"].max())
        idx = torch.arange(w, device="cuda").view(1, -1) < qa["produced"].view(-1, 1)
        assert torch.equal(qa["bits"][:, :w][idx], qb["bits"][:, :w][idx])
    assert a.preamble_detect.tags().tobytes() == b.preamble_detect.tags().tobytes()


def test_time_parallel_pipelined_symbols_only(ais):
    """aisx_msk_process_stream_after with a ready event and NO bit tail (d_bits NULL: the gather of the units'
    symbols runs on the caller's stream): six calls queued back to back.  The units of call k + 2 reuse the
    staging rows call k's gather reads, and run on the handle's own stream -- they must wait for that gather
    (the event they wait for is recorded behind it).  Symbols and counts equal to the serial kernel's."""
    import ctypes as C
    import torch
    import synth
    from ais_amd import _lib

    nchan, L, ncalls, sps, Q = 512, 32768, 6, 4.0, 256
    rng = np.random.default_rng(11)
    total = L * ncalls
    base = np.stack([synth.make_channel(4000 + c, total, "P", 4, amp=1.0, cfo_max=50.0)[0] for c in range(8)])
    xs = np.concatenate([np.roll(base, 1000 * r, axis=1) for r in range(nchan // 8)], axis=0)
    tags8 = [_tags_with_pairs(rng, total, c, sps, 700, 0.2, 6, None) for c in range(8)]
    cap = max(len(t) for t in tags8) + 1
    d_x = _dev(xs)
    calls = []
    for k in range(ncalls):
        tg = np.zeros((nchan, cap), dtype=ais.TAG_DTYPE)
        cnt = np.zeros(nchan, np.int32)
        for c in range(nchan):
            t = tags8[c % 8]
            sel = t[(t["offset"] >= k * L) & (t["offset"] < (k + 1) * L)]
            for f in ("offset", "value", "key"):
                tg[f][c, : len(sel)] = sel[f]
            tg["chan"][c, : len(sel)] = c
            cnt[c] = len(sel)
        calls.append((torch.as_tensor(tg.view(np.uint8).reshape(nchan, -1).copy()).cuda(), torch.as_tensor(cnt).cuda()))

    def run(time_parallel):
        blk = ais.msk_timing_recovery_cc(sps, 0.04, 0.01, 1, nchan=nchan, max_items=L)
        blk.set_max_noutput_items(Q)
        if time_parallel:
            blk.set_time_parallel(64, join_kernel=1, max_unit_items=16384)
        ocap = blk.out_capacity
        st = torch.cuda.Stream()
        outs = []
        with torch.cuda.stream(st):
            for k in range(ncalls):
                syms = torch.zeros((nchan, ocap), dtype=torch.complex64, device="cuda")
                prod = torch.zeros(nchan, dtype=torch.int32, device="cuda")
                ev = torch.cuda.Event()
                ev.record(st)
                x = d_x[:, k * L:(k + 1) * L]
                rc = _lib.lib().aisx_msk_process_stream_after(
                    blk._h, x.data_ptr(), x.stride(0), L, calls[k][0].data_ptr(), calls[k][1].data_ptr(), cap,
                    syms.data_ptr(), None, None, None, ocap, prod.data_ptr(), C.c_void_p(st.cuda_stream),
                    C.c_void_p(ev.cuda_event))
                _lib.check(rc, "process_stream_after")
                outs.append((syms, prod, ev))
        st.synchronize()
        assert blk.last_status() == 0
        stats = blk.restart_stats() if time_parallel else None
        return [(s.cpu().numpy(), p.cpu().numpy()) for s, p, _ in outs], stats

    want, _ = run(False)
    for trial in range(2):
        got, stats = run(True)
        assert stats["units_taken"] > 0
        for k in range(ncalls):
            assert np.array_equal(got[k][1], want[k][1]), (trial, k)
            p = want[k][1]
            for c in range(nchan):
                assert np.array_equal(got[k][0][c, :p[c]].view(np.uint32), want[k][0][c, :p[c]].view(np.uint32)), (trial, k, c)
