"""BASELINE config 1 -- one 4-sps channel through the stock ais_demod.py flowgraph -- driven the way
the GNU Radio scheduler drives the blocks, WITHOUT GNU Radio: a small deterministic scheduler
model shared by the fixture generator (oracle side), by the Python test and, restated in C++, by
tests/abi_cpp/sched_harness.cpp (which calls libaisx.so's *_work_host entry points only).

The policy (what a scheduler is free to choose, pinned here so that both sides make the same
choices):
  source      hands freq_sync SRC_PIECES[i % len] new items per call;
  freq_sync   one call per piece (square_and_fft_sync_cc: whole fftlen-vectors come out);
  agc         sync_block, history 512: one call over everything that has arrived;
  corr_est    sync_block, history N + 1, output multiple m = output_multiple(), at most 24576 items
              (lib/corr_est_cc_impl.cc:84-85,95,111-112): noutput = min(avail // m, CORR_K[j % len],
              24576 // m) * m, nothing if that is 0;
  msk         general block (lib/msk_timing_recovery_cc_impl.cc:98-105): noutput starts at
              MSK_CAPS[k % len] and is halved until forecast(noutput) <= items available (the
              scheduler's own back-off), nothing if it reaches 0; ninput_items = everything
              available; the item one past ninput_items reads as zero (in_has_lookahead = 0);
              tags = the time_est tags of the store at or after nitems_read.
"""
import numpy as np

SRC_PIECES = [4096, 1000, 24, 5000, 7168]
CORR_K = [3, 21, 1, 8]
MSK_CAPS = [512, 100, 1, 700, 2048]
MAX_NOUTPUT = 24 * 1024
AGC_W = 512


class Blocks:
    """What the scheduler model needs from an implementation (oracle or libaisx host path)."""

    def freqsync(self, x): ...
    def agc(self, buf_with_history, n): ...
    def corr(self, buf_with_history, n, written): ...  # -> out, tags(TAG rows: offset,value,key)
    def corr_geometry(self): ...                       # -> (N, output_multiple)
    def msk_forecast(self, nout): ...
    def msk(self, nout, ninput, items, tags, nitems_read): ...  # -> bits, consumed


def run(blocks, x):
    N, m = blocks.corr_geometry()
    agc_hist = np.zeros(AGC_W - 1, np.complex64)
    corr_hist = np.zeros(N, np.complex64)
    corr_pending = np.zeros(0, np.complex64)
    corr_written = 0
    msk_buf = np.zeros(0, np.complex64)
    msk_read = 0
    store = []  # (offset, value, key) of every corr_est tag, in emission order
    bits = []
    calls = []
    pos = si = cj = mk = 0
    while pos < len(x):
        piece = x[pos:pos + SRC_PIECES[si % len(SRC_PIECES)]]
        si += 1
        pos += len(piece)
        y1 = blocks.freqsync(piece)
        if len(y1) == 0:
            continue
        y2 = blocks.agc(np.concatenate([agc_hist, y1]), len(y1))
        agc_hist = np.concatenate([agc_hist, y1])[len(y1):]
        corr_pending = np.concatenate([corr_pending, y2])
        while True:
            k = min(len(corr_pending) // m, CORR_K[cj % len(CORR_K)], MAX_NOUTPUT // m)
            if k == 0:
                break
            cj += 1
            n = k * m
            out, tags = blocks.corr(np.concatenate([corr_hist, corr_pending[:n]]), n, corr_written)
            corr_hist = np.concatenate([corr_hist, corr_pending[:n]])[n:]
            corr_pending = corr_pending[n:]
            corr_written += n
            store.extend((int(t["offset"]), float(t["value"]), int(t["key"])) for t in tags)
            msk_buf = np.concatenate([msk_buf, out])
            while True:
                avail = len(msk_buf)
                nout = MSK_CAPS[mk % len(MSK_CAPS)]
                while nout > 0 and blocks.msk_forecast(nout) > avail:
                    nout >>= 1
                if nout == 0:
                    break
                mk += 1
                live = [t for t in store if t[2] == 2 and t[0] >= msk_read]
                b, consumed = blocks.msk(nout, avail, msk_buf, live, msk_read)
                calls.append((nout, avail, consumed, len(b)))
                bits.append(np.asarray(b, np.uint8))
                msk_buf = msk_buf[consumed:]
                msk_read += consumed
                if consumed == 0 and len(b) == 0:
                    break
    return dict(bits=np.concatenate(bits) if bits else np.zeros(0, np.uint8), tags=store, calls=calls)


class OracleBlocks(Blocks):
    def __init__(self, tmpl, sps=4.0):
        import oracle_py as orc

        self.orc = orc
        self.fs = orc.FreqSync(9600.0 * sps, 9600.0, 1024)
        self.ce = orc.CorrEst(tmpl, sps, 1, 0.9)
        self.mskb = orc.Msk(sps, 0.04, 0.01, 1)
        self.bt = orc.BitTail()

    def freqsync(self, x):
        return self.fs.process(x)[0]

    def agc(self, buf, n):
        import ctypes as C

        out = np.zeros(n, np.complex64)
        b = np.ascontiguousarray(buf, np.complex64)
        self.orc.lib().orc_feedforward_agc(AGC_W, 2.0, n, b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def corr_geometry(self):
        return self.ce.N, self.ce.output_multiple

    def corr(self, buf, n, written):
        # (CorrEst keeps its own copy of the history: it is the same N items)
        assert np.array_equal(np.asarray(buf[:self.ce.N]), self.ce.hist) and written == self.ce.written
        out, _, tags = self.ce.work(buf[self.ce.N:self.ce.N + n])
        return out, tags

    def msk_forecast(self, nout):
        return self.mskb.forecast(nout)

    def msk(self, nout, ninput, items, tags, nitems_read):
        orc = self.orc
        t = np.zeros(len(tags), dtype=orc.TAG_DTYPE)
        for i, (o, v, k) in enumerate(tags):
            t[i]["offset"], t[i]["value"], t[i]["key"] = o, v, k
        # in[-1] is addressable (the item before nitems_read; zero at the stream start is what the
        # harness keeps too), the item one past ninput_items reads as zero
        buf = np.concatenate([self._prev(), items[:ninput], np.zeros(8, np.complex64)])
        out, _, _, cons, st = self.mskb.general_work(nout, ninput, buf, 1, t, nitems_read)
        if cons > 0:
            self._last = items[cons - 1]
        return self.bt.process(out), cons

    def _prev(self):
        return np.array([getattr(self, "_last", 0)], np.complex64)


def read_fixture(path):
    """tests/golden/config1_sched.bin (layout: tests/golden/make_config1.py)."""
    import struct

    with open(path, "rb") as f:
        assert f.read(8) == b"AISXC1\0\0"
        N, T, ntags, nbits, nbursts, nsrc, nck, ncaps = struct.unpack("<8i", f.read(32))
        (sps,) = struct.unpack("<f", f.read(4))
        tmpl = np.frombuffer(f.read(8 * N), "<c8")
        x = np.frombuffer(f.read(8 * T), "<c8")
        src = np.frombuffer(f.read(4 * nsrc), "<i4")
        ck = np.frombuffer(f.read(4 * nck), "<i4")
        caps = np.frombuffer(f.read(4 * ncaps), "<i4")
        tags = np.frombuffer(f.read(24 * ntags), dtype=[("offset", "<u8"), ("value", "<f8"), ("key", "<i4"), ("chan", "<i4")])
        bits = np.frombuffer(f.read(nbits), np.uint8)
        bursts = np.frombuffer(f.read(8 * nbursts), "<i4").reshape(-1, 2)
    assert list(src) == SRC_PIECES and list(ck) == CORR_K and list(caps) == MSK_CAPS
    return dict(sps=sps, tmpl=tmpl, x=x, tags=tags, bits=bits, bursts=bursts)
