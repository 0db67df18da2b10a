"""ctypes binding of the CPU oracle (oracle/libais_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never imported by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

KEY_CORR_START, KEY_PHASE_EST, KEY_TIME_EST, KEY_CORR_EST = 0, 1, 2, 3
KEY_NAMES = {0: "corr_start", 1: "phase_est", 2: "time_est", 3: "corr_est"}

TAG_DTYPE = np.dtype([("offset", "<u8"), ("value", "<f8"), ("key", "<i4"), ("port", "<i4")])


def build():
    so = os.path.join(ORACLE_DIR, "libais_oracle.so")
    src = [os.path.join(ORACLE_DIR, f) for f in ("ais_oracle.c", "ais_oracle.h", "orc_tables.h")]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _bind(C.CDLL(build()))
    return _LIB


class use_library:
    """with use_library(path): every oracle_py object created AND used inside talks to the oracle built
    at `path` instead (tests/test_table_sensitivity.py: the same C file compiled against perturbed tables)."""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        global _LIB
        lib()
        self.saved = _LIB
        _LIB = _bind(C.CDLL(self.path))
        return _LIB

    def __exit__(self, *exc):
        global _LIB
        _LIB = self.saved
        return False


def _bind(L):
    if True:
        vp, i32, u32, f32, f64, u64 = C.c_void_p, C.c_int, C.c_uint, C.c_float, C.c_double, C.c_uint64
        pi32 = C.POINTER(C.c_int)
        L.orc_fast_atan2f.restype = f32
        L.orc_fast_atan2f.argtypes = [f32, f32]
        L.orc_branchless_clip.restype = f32
        L.orc_branchless_clip.argtypes = [f32, f32]
        L.orc_nco_sincos.argtypes = [f32, vp, vp]
        L.orc_fxpt_float_to_fixed.restype = C.c_int32
        L.orc_fxpt_float_to_fixed.argtypes = [f32]
        L.orc_fft.argtypes = [vp, i32, i32]
        L.orc_corr_create.restype = vp
        L.orc_corr_create.argtypes = [vp, i32, f32, u32, f32]
        L.orc_corr_destroy.argtypes = [vp]
        for n in ("orc_corr_history", "orc_corr_output_multiple", "orc_corr_fftsize"):
            getattr(L, n).restype = i32
            getattr(L, n).argtypes = [vp]
        L.orc_corr_threshold.restype = f32
        L.orc_corr_threshold.argtypes = [vp]
        L.orc_corr_mark_delay.restype = u32
        L.orc_corr_mark_delay.argtypes = [vp]
        L.orc_corr_taps.argtypes = [vp, vp]
        L.orc_corr_set_symbols.argtypes = [vp, vp, i32]
        L.orc_corr_work.restype = i32
        L.orc_corr_work.argtypes = [vp, i32, vp, vp, vp, u64, vp, i32, pi32]
        L.orc_freqest_init.argtypes = [vp, f32, i32, i32]
        L.orc_freqest_work.restype = i32
        L.orc_freqest_work.argtypes = [vp, i32, vp, vp]
        L.orc_freqsync_create.restype = vp
        L.orc_freqsync_create.argtypes = [f64, f64, i32]
        L.orc_freqsync_destroy.argtypes = [vp]
        L.orc_freqsync_process.restype = i32
        L.orc_freqsync_process.argtypes = [vp, vp, i32, vp, vp]
        L.orc_feedforward_agc.argtypes = [i32, f32, i32, vp, vp]
        L.orc_feedforward_agc_floor.argtypes = [i32, f32, f32, i32, vp, vp]
        L.orc_msk_create.restype = vp
        L.orc_msk_create.argtypes = [f32, f32, f32, i32, pi32]
        L.orc_msk_destroy.argtypes = [vp]
        L.orc_msk_set_gain.restype = i32
        L.orc_msk_set_gain.argtypes = [vp, f32]
        L.orc_msk_get_gain.restype = f32
        L.orc_msk_get_gain.argtypes = [vp]
        L.orc_msk_set_limit.argtypes = [vp, f32]
        L.orc_msk_get_limit.restype = f32
        L.orc_msk_get_limit.argtypes = [vp]
        L.orc_msk_set_sps.argtypes = [vp, f32]
        L.orc_msk_get_sps.restype = f32
        L.orc_msk_get_sps.argtypes = [vp]
        L.orc_msk_forecast.restype = i32
        L.orc_msk_forecast.argtypes = [vp, i32]
        L.orc_msk_general_work.restype = i32
        L.orc_msk_general_work.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, i32, u64, pi32, pi32]
        L.orc_msk_get_state.argtypes = [vp, vp, pi32]
        L.orc_bittail_init.argtypes = [vp]
        L.orc_bittail_process.argtypes = [vp, vp, i32, vp]
        L.orc_gmsk_modulate_vector.restype = i32
        L.orc_gmsk_modulate_vector.argtypes = [i32, f64, vp, i32, vp]
        L.orc_demod_create.restype = vp
        L.orc_demod_create.argtypes = [f32, f32, f32, f32, i32, vp, i32, i32]
        L.orc_demod_destroy.argtypes = [vp]
        L.orc_demod_step.restype = i32
        L.orc_demod_step.argtypes = [vp, vp, i32, vp, i32, vp, vp, i32, pi32]
        L.orc_demod_set_max_noutput.argtypes = [vp, i32]
        L.orc_demod_bench_mt.restype = C.c_long
        L.orc_demod_bench_mt.argtypes = [i32, f32, vp, i32, i32, vp, i32, i32, f64, C.POINTER(C.c_double)]
        L.orc_demod_hash.restype = u64
        L.orc_demod_hash.argtypes = [f32, vp, i32, i32, vp, i32]
        L.orc_hdlc_init.argtypes = [vp, i32, i32]
        L.orc_hdlc_work.restype = i32
        L.orc_hdlc_work.argtypes = [vp, vp, i32, vp, i32, vp, i32]
        L.orc_pdu_to_nmea.restype = i32
        L.orc_pdu_to_nmea.argtypes = [C.c_char_p, vp, i32, C.c_char_p, i32]
        L.orc_freq_xlating_fir.argtypes = [vp, i32, i32, f64, f64, vp, C.c_long, C.c_long, i32, vp]
        L.orc_firdes_low_pass.restype = i32
        L.orc_firdes_low_pass.argtypes = [f64, f64, f64, f64, vp, i32]
    return L


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.complex64)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def fast_atan2f(y, x):
    return float(lib().orc_fast_atan2f(y, x))


def nco_sincos(phase):
    """[GR] frequency_modulator_fc's sin / cos of d_phase (gr::fxpt)"""
    s = C.c_float()
    c = C.c_float()
    lib().orc_nco_sincos(phase, C.byref(s), C.byref(c))
    return s.value, c.value


def fft(x, inverse=False):
    a = _c64(x).copy()
    lib().orc_fft(_ptr(a), a.size, 1 if inverse else 0)
    return a


def gmsk_modulate_vector(sps, bt, data):
    d = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.zeros(d.size * 8 * sps, dtype=np.complex64)
    n = lib().orc_gmsk_modulate_vector(int(sps), float(bt), _ptr(d), d.size, _ptr(out))
    return out[:n]


class CorrEst:
    """corr_est_cc driven like a GNU Radio sync_block: keeps the history."""

    def __init__(self, symbols, sps, mark_delay, threshold=0.9):
        s = _c64(symbols)
        self.h = lib().orc_corr_create(_ptr(s), s.size, sps, mark_delay, threshold)
        self.N = s.size
        self.hist = np.zeros(self.N, dtype=np.complex64)
        self.written = 0

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_corr_destroy(self.h)
            self.h = None

    history = property(lambda self: lib().orc_corr_history(self.h))
    output_multiple = property(lambda self: lib().orc_corr_output_multiple(self.h))
    fftsize = property(lambda self: lib().orc_corr_fftsize(self.h))
    threshold = property(lambda self: lib().orc_corr_threshold(self.h))
    mark_delay = property(lambda self: lib().orc_corr_mark_delay(self.h))

    def taps(self):
        out = np.zeros(self.N, dtype=np.complex64)
        lib().orc_corr_taps(self.h, _ptr(out))
        return out

    def set_symbols(self, symbols):
        """set_symbols() (lib/corr_est_cc_impl.cc:132-162).  The scheduler's history follows
        set_history(N + 1): the N items before the next new one (zeros before the stream start)."""
        s = _c64(symbols)
        lib().orc_corr_set_symbols(self.h, _ptr(s), s.size)
        keep = min(self.N, s.size)
        hist = np.zeros(s.size, dtype=np.complex64)
        if keep:
            hist[s.size - keep:] = self.hist[self.N - keep:]
        self.hist, self.N = hist, s.size

    def work(self, x, want_corr=False):
        """One work() call on len(x) new items.  Returns (out, corr|None, tags)."""
        x = _c64(x)
        n = x.size
        buf = np.concatenate([self.hist, x])
        out = np.zeros(n, dtype=np.complex64)
        corr = np.zeros(n, dtype=np.complex64) if want_corr else None
        maxt = 7 * (n + 1)
        tags = np.zeros(maxt, dtype=TAG_DTYPE)
        nt = C.c_int(0)
        lib().orc_corr_work(self.h, n, _ptr(buf), _ptr(out), _ptr(corr) if want_corr else None,
                            self.written, _ptr(tags), maxt, C.byref(nt))
        self.hist = buf[n:].copy()
        self.written += n
        return out, corr, tags[: nt.value].copy()


class FreqEst(C.Structure):
    _fields_ = [("binsize", C.c_float), ("offset", C.c_int), ("fftlen", C.c_int)]

    @classmethod
    def make(cls, sample_rate, data_rate, fftlen):
        f = cls()
        lib().orc_freqest_init(C.byref(f), sample_rate, data_rate, fftlen)
        return f

    def work(self, vecs):
        v = _c64(vecs).reshape(-1, self.fftlen)
        out = np.zeros(v.shape[0], dtype=np.float32)
        lib().orc_freqest_work(C.byref(self), v.shape[0], _ptr(v), _ptr(out))
        return out


class FreqSync:
    def __init__(self, samplerate, bits_per_sec, fftlen):
        self.h = lib().orc_freqsync_create(samplerate, bits_per_sec, fftlen)
        self.fftlen = fftlen

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_freqsync_destroy(self.h)
            self.h = None

    def process(self, x):
        x = _c64(x)
        out = np.zeros(x.size + self.fftlen, dtype=np.complex64)
        fh = np.zeros(x.size // self.fftlen + 2, dtype=np.float32)
        n = lib().orc_freqsync_process(self.h, _ptr(x), x.size, _ptr(out), _ptr(fh))
        return out[:n].copy(), fh[: n // self.fftlen].copy()


class Agc:
    def __init__(self, nsamples=512, reference=2.0, floor=None):
        self.ns, self.ref, self.floor = nsamples, reference, floor
        self.hist = np.zeros(nsamples - 1, dtype=np.complex64)

    def work(self, x):
        x = _c64(x)
        buf = np.concatenate([self.hist, x])
        out = np.zeros(x.size, dtype=np.complex64)
        if self.floor is None:
            lib().orc_feedforward_agc(self.ns, self.ref, x.size, _ptr(buf), _ptr(out))
        else:
            lib().orc_feedforward_agc_floor(self.ns, self.ref, self.floor, x.size, _ptr(buf), _ptr(out))
        self.hist = buf[x.size:].copy()
        return out


class Msk:
    def __init__(self, sps, gain, limit, osps=1):
        err = C.c_int(0)
        self.h = lib().orc_msk_create(sps, gain, limit, osps, C.byref(err))
        if not self.h:
            raise IndexError("out_of_range (%d)" % err.value)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_msk_destroy(self.h)
            self.h = None

    def forecast(self, nout):
        return lib().orc_msk_forecast(self.h, nout)

    def set_gain(self, g):
        if lib().orc_msk_set_gain(self.h, g) != 0:
            raise IndexError("Gain must be positive")

    def state(self):
        s = np.zeros(8, dtype=np.float32)
        d = C.c_int(0)
        lib().orc_msk_get_state(self.h, _ptr(s), C.byref(d))
        return s, d.value

    def general_work(self, noutput, ninput, buf, in_off, tags, nitems_read, want_aux=False):
        """buf: complex64 array; in = &buf[in_off] (so in[-1] is addressable)."""
        buf = _c64(buf)
        out = np.zeros(max(noutput, 1), dtype=np.complex64)
        o2 = np.zeros(max(noutput, 1), dtype=np.float32) if want_aux else None
        o3 = np.zeros(max(noutput, 1), dtype=np.float32) if want_aux else None
        tags = np.ascontiguousarray(tags, dtype=TAG_DTYPE)
        cons = C.c_int(0)
        st = C.c_int(0)
        inp = C.c_void_p(buf.ctypes.data + 8 * in_off)
        prod = lib().orc_msk_general_work(self.h, noutput, ninput, inp, _ptr(out),
                                          _ptr(o2) if want_aux else None, _ptr(o3) if want_aux else None,
                                          _ptr(tags), tags.size, nitems_read, C.byref(cons), C.byref(st))
        return out[:prod].copy(), (o2[:prod].copy() if want_aux else None), (o3[:prod].copy() if want_aux else None), cons.value, st.value


class MskStream:
    """msk_timing_recovery_cc under the step contract of orc_demod_step /
    aisx chain: all pending input minus one look-ahead item is offered, and
    noutput_items is the largest count whose forecast() fits."""

    def __init__(self, sps, gain, limit, osps=1, max_noutput=0):
        self.m = Msk(sps, gain, limit, osps)
        self.max_noutput = max_noutput  # gr::block::set_max_noutput_items() (0: whatever fits)
        self.buf = np.zeros(1, dtype=np.complex64)  # [0] = item before nitems_read
        self.read = 0
        self.store = np.zeros(0, dtype=TAG_DTYPE)

    def step(self, x, new_tags, want_aux=False):
        x = _c64(x)
        self.buf = np.concatenate([self.buf, x])
        if len(new_tags):
            self.store = np.concatenate([self.store, np.asarray(new_tags, dtype=TAG_DTYPE)])
            self.store = self.store[np.argsort(self.store["offset"], kind="stable")]
        outs, o2s, o3s, total = [], [], [], 0
        while True:  # the scheduler keeps calling general_work until forecast(1) no longer fits
            pending = self.buf.size - 1
            ninput = pending - 1
            nout = 0
            dsps = lib().orc_msk_get_sps(self.m.h)
            if ninput > 0:
                nout = int((ninput - 3.0 * dsps - 8) / (2.0 * dsps)) + 2
                while nout > 0 and self.m.forecast(nout) > ninput:
                    nout -= 1
            if self.max_noutput > 0:
                nout = min(nout, self.max_noutput)
            if nout <= 0 or int(ninput - 3.0 * dsps) <= 0:
                break
            # items past the ones on offer read as zero (the reference may look a few items
            # past ninput_items when sps < 4)
            padded = np.concatenate([self.buf, np.zeros(8, np.complex64)])
            out, o2, o3, cons, st = self.m.general_work(nout, ninput, padded, 1, self.store, self.read, want_aux)
            self.buf = self.buf[cons:].copy()
            self.read += cons
            self.store = self.store[self.store["offset"] >= self.read]
            outs.append(out)
            o2s.append(o2)
            o3s.append(o3)
            total += cons
            if cons <= 0:  # (nothing consumed: the step ends, see orc_demod_step)
                break
        if not outs:
            return np.zeros(0, np.complex64), np.zeros(0, np.float32), np.zeros(0, np.float32), 0
        cat = np.concatenate
        return cat(outs), (cat(o2s) if want_aux else None), (cat(o3s) if want_aux else None), total


class BitTail(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float), ("prev_bit", C.c_ubyte)]

    def __init__(self):
        super().__init__()
        lib().orc_bittail_init(C.byref(self))

    def process(self, syms):
        s = _c64(syms)
        bits = np.zeros(s.size, dtype=np.uint8)
        lib().orc_bittail_process(C.byref(self), _ptr(s), s.size, _ptr(bits))
        return bits


class Demod:
    """python/ais_demod.py chain for one channel (orc_demod_*)."""

    def __init__(self, sps, symbols, gain=0.04, limit=0.01, fftlen=1024, bits_per_sec=9600.0, stages=3, max_noutput=0):
        s = _c64(symbols)
        self.h = lib().orc_demod_create(sps, bits_per_sec, gain, limit, fftlen, _ptr(s), s.size, stages)
        if max_noutput:
            lib().orc_demod_set_max_noutput(self.h, max_noutput)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_demod_destroy(self.h)
            self.h = None

    def step(self, x, want_syms=False):
        x = _c64(x)
        maxb = x.size + 64
        bits = np.zeros(maxb, dtype=np.uint8)
        syms = np.zeros(maxb, dtype=np.complex64) if want_syms else None
        maxt = 4 * x.size + 16
        tags = np.zeros(maxt, dtype=TAG_DTYPE)
        nt = C.c_int(0)
        nb = lib().orc_demod_step(self.h, _ptr(x), x.size, _ptr(bits), maxb, _ptr(syms) if want_syms else None,
                                  _ptr(tags), maxt, C.byref(nt))
        return bits[:nb].copy(), (syms[:nb].copy() if want_syms else None), tags[: min(nt.value, maxt)].copy()


def build_native(out_path):
    """The CPU-baseline build of the same C file, compiled on THIS host with -O3 -march=native
    (oracle/Makefile `native`); returns a ctypes handle with the timing entry points, or None if
    the host has no working compiler."""
    try:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "native", "OUT=%s" % out_path],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        L = C.CDLL(out_path)
    except (OSError, subprocess.CalledProcessError):
        return None
    vp, i32, f32, f64, u64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_uint64
    L.orc_demod_bench_mt.restype = C.c_long
    L.orc_demod_bench_mt.argtypes = [i32, f32, vp, i32, i32, vp, i32, i32, f64, C.POINTER(C.c_double)]
    L.orc_demod_hash.restype = u64
    L.orc_demod_hash.argtypes = [f32, vp, i32, i32, vp, i32]
    return L


def demod_hash(sps, symbols, stages, x, L=None):
    """FNV-1a over the bits and tags of one channel through a fresh chain (orc_demod_hash)"""
    s, a = _c64(symbols), _c64(x)
    return int((L or lib()).orc_demod_hash(float(sps), _ptr(s), s.size, int(stages), _ptr(a), a.size))


def demod_bench_mt(nthreads, sps, symbols, stages, xs, budget_s, L=None):
    """CPU baseline: (channels completed, wall seconds) of `nthreads` C threads running whole
    channels (rows of xs) through the oracle chain for ~budget_s seconds."""
    s = _c64(symbols)
    x = np.ascontiguousarray(xs, dtype=np.complex64)
    wall = C.c_double(0)
    done = (L or lib()).orc_demod_bench_mt(int(nthreads), float(sps), _ptr(s), s.size, int(stages), _ptr(x), x.shape[1],
                                           x.shape[0], float(budget_s), C.byref(wall))
    return int(done), wall.value


class Hdlc(C.Structure):
    _fields_ = [("length_min", C.c_int), ("length_max", C.c_int), ("ones", C.c_int), ("bitctr", C.c_int),
                ("bytectr", C.c_int), ("pktbuf", C.c_ubyte * 1024)]

    def __init__(self, length_min=11, length_max=64):
        super().__init__()
        lib().orc_hdlc_init(C.byref(self), length_min, length_max)

    def work(self, bits):
        b = np.ascontiguousarray(bits, dtype=np.uint8)
        maxf = b.size // 16 + 2
        out = np.zeros(maxf * 66, dtype=np.uint8)
        offs = np.zeros(maxf + 1, dtype=np.int32)
        n = lib().orc_hdlc_work(C.byref(self), _ptr(b), b.size, _ptr(out), out.size, _ptr(offs), maxf)
        return [bytes(out[offs[k]:offs[k + 1]]) for k in range(n)]


def pdu_to_nmea(designator, pdu):
    p = np.frombuffer(bytes(pdu), dtype=np.uint8)
    out = C.create_string_buffer(4096)
    n = lib().orc_pdu_to_nmea(designator.encode(), _ptr(p), p.size, out, 4096)
    return out.raw[:n].decode("latin-1")


def firdes_low_pass(gain, fs, cutoff, transition):
    cap = int(53.0 * fs / (22.0 * transition)) + 4
    t = np.zeros(cap, dtype=np.float32)
    n = lib().orc_firdes_low_pass(gain, fs, cutoff, transition, _ptr(t), cap)
    return t[:n].copy()


def freq_xlating_fir(taps, decim, center_freq, fs, x, k0, nout):
    t = np.ascontiguousarray(taps, dtype=np.float32)
    x = _c64(x)
    out = np.zeros(nout, dtype=np.complex64)
    lib().orc_freq_xlating_fir(_ptr(t), t.size, decim, center_freq, fs, _ptr(x), x.size, k0, nout, _ptr(out))
    return out
