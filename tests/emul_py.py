"""ctypes binding of tests/emul/libaisx_emul.so: the CPU model that runs the
product's kernel-body templates (gr-ais_amd/csrc/k_*.h) one OS thread per lane.
TEST INFRASTRUCTURE for the -m "not gpu" suite."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
EMUL_DIR = os.path.join(_HERE, "emul")
CSRC = os.path.join(os.path.dirname(_HERE), "gr-ais_amd", "csrc")
_LIB = None

TAG_DTYPE = np.dtype([("offset", "<u8"), ("value", "<f8"), ("key", "<i4"), ("chan", "<i4")])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(EMUL_DIR, "libaisx_emul.so")
        deps = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(EMUL_DIR, "emul.cpp")]
        if (not os.path.exists(so)) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            subprocess.check_call(["make", "-C", EMUL_DIR, "-s", "-B"])
        L = C.CDLL(so)
        vp, i32, f32, u32, u64, lng = C.c_void_p, C.c_int, C.c_float, C.c_uint, C.c_uint64, C.c_long
        L.emu_corr_create.restype = vp
        L.emu_corr_create.argtypes = [vp, i32, f32, u32, f32, i32]
        L.emu_corr_destroy.argtypes = [vp]
        L.emu_corr_set_dma.argtypes = [i32]
        L.emu_corr_threshold.restype = f32
        L.emu_corr_threshold.argtypes = [vp]
        L.emu_corr_output_multiple.restype = i32
        L.emu_corr_output_multiple.argtypes = [vp]
        L.emu_corr_symbols.argtypes = [vp, vp]
        L.emu_corr_process.restype = i32
        L.emu_corr_process.argtypes = [vp, vp, lng, vp, lng, vp, lng, i32, vp, i32, vp, i32]
        L.emu_msk_create.restype = vp
        L.emu_msk_create.argtypes = [f32, f32, f32, i32, i32]
        L.emu_msk_destroy.argtypes = [vp]
        L.emu_msk_set_lpw.argtypes = [vp, i32]
        L.emu_msk_set_time_parallel.argtypes = [vp, i32, i32, i32]
        L.emu_msk_tp_stats.argtypes = [vp, vp]
        L.emu_msk_set_tp_join.argtypes = [vp, i32]
        L.emu_msk_process_stream.restype = i32
        L.emu_msk_process_stream.argtypes = [vp, vp, lng, i32, vp, vp, i32, vp, vp, vp, vp, lng, vp, vp]
        L.emu_msk_general_work.restype = i32
        L.emu_msk_general_work.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, u64, vp, vp]
        f64 = C.c_double
        L.emu_agc_create.restype = vp
        L.emu_agc_create.argtypes = [i32, f32, i32]
        L.emu_agc_destroy.argtypes = [vp]
        L.emu_agc_process.argtypes = [vp, vp, lng, vp, lng, i32]
        L.emu_agc_set_streaming.argtypes = [vp, i32]
        L.emu_fs_agc_process.restype = i32
        L.emu_fs_agc_process.argtypes = [vp, vp, vp, lng, i32, vp, lng, vp, lng]
        L.emu_fs_create.restype = vp
        L.emu_fs_create.argtypes = [f64, f64, i32, i32, i32]
        L.emu_fs_destroy.argtypes = [vp]
        L.emu_fs_process.restype = i32
        L.emu_fs_process.argtypes = [vp, vp, lng, i32, vp, lng, vp, lng]
        L.emu_freqest_work.argtypes = [vp, vp, lng, vp, lng, i32]
        L.emu_freqest_any.argtypes = [vp, lng, vp, lng, i32, i32, i32, f32, i32]
        L.emu_pfb_create.restype = vp
        L.emu_pfb_create.argtypes = [i32, vp, i32, i32]
        L.emu_pfb_destroy.argtypes = [vp]
        L.emu_pfb_process.restype = i32
        L.emu_pfb_process.argtypes = [vp, vp, lng, i32, vp, lng]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class CorrEst:
    def __init__(self, symbols, sps, mark_delay, threshold=0.9, nchan=1):
        s = np.ascontiguousarray(symbols, dtype=np.complex64)
        self.h = lib().emu_corr_create(_p(s), s.size, sps, mark_delay, threshold, nchan)
        self.N, self.nchan = s.size, nchan

    def __del__(self):
        if getattr(self, "h", None):
            lib().emu_corr_destroy(self.h)
            self.h = None

    threshold = property(lambda self: lib().emu_corr_threshold(self.h))
    output_multiple = property(lambda self: lib().emu_corr_output_multiple(self.h))

    def symbols(self):
        out = np.zeros(self.N, np.complex64)
        lib().emu_corr_symbols(self.h, _p(out))
        return out

    def work(self, x, want_corr=False, tag_cap=256, force_nseg=0):
        """x: [nchan][n] complex64.  Returns out, corr|None, list of per-channel tag arrays."""
        x = np.ascontiguousarray(x, dtype=np.complex64).reshape(self.nchan, -1)
        n = x.shape[1]
        out = np.zeros_like(x)
        corr = np.zeros_like(x) if want_corr else None
        tags = np.zeros((self.nchan, tag_cap), dtype=TAG_DTYPE)
        cnt = np.zeros(self.nchan, dtype=np.int32)
        lib().emu_corr_process(self.h, _p(x), n, _p(out), n, _p(corr), n, n, _p(tags), tag_cap, _p(cnt), force_nseg)
        return out, corr, [tags[c, : min(cnt[c], tag_cap)].copy() for c in range(self.nchan)], cnt, tags


class MskStream:
    def __init__(self, sps, gain, limit, osps=1, nchan=1, lpw=64, tp_smax=-1, tp_min_gap=256, max_noutput=0, tp_join=0):
        self.h = lib().emu_msk_create(sps, gain, limit, osps, nchan)
        self.nchan = nchan
        lib().emu_msk_set_lpw(self.h, lpw)  # channels per wave: 16, 32 or 64
        # tp_smax >= 0: the time-parallel kernels (k_mskp.h) with that many restart points per channel at most
        lib().emu_msk_set_time_parallel(self.h, tp_smax, tp_min_gap, max_noutput)
        lib().emu_msk_set_tp_join(self.h, tp_join)  # 0: the lane-per-channel join of k_mskp.h, 1: the serial kernel with fast-forward

    def tp_stats(self):
        a = np.zeros(4, np.int64)
        lib().emu_msk_tp_stats(self.h, _p(a))
        return dict(restart_points=int(a[0]), units_accepted=int(a[1]), symbols_from_units=int(a[2]), symbols=int(a[3]))

    def __del__(self):
        if getattr(self, "h", None):
            lib().emu_msk_destroy(self.h)
            self.h = None

    def step(self, x, tags=None, tag_counts=None, want_aux=False, out_cap=None):
        x = np.ascontiguousarray(x, dtype=np.complex64).reshape(self.nchan, -1)
        n = x.shape[1]
        cap = out_cap or (n // 2 + 300)
        syms = np.zeros((self.nchan, cap), np.complex64)
        bits = np.zeros((self.nchan, cap), np.uint8)
        err = np.zeros((self.nchan, cap), np.float32) if want_aux else None
        mu = np.zeros((self.nchan, cap), np.float32) if want_aux else None
        prod = np.zeros(self.nchan, np.int32)
        cons = np.zeros(self.nchan, np.int32)
        if tags is not None:
            tags = np.ascontiguousarray(tags, dtype=TAG_DTYPE).reshape(self.nchan, -1)
            tag_counts = np.ascontiguousarray(tag_counts, dtype=np.int32)
            tcap = tags.shape[1]
        else:
            tcap = 0
        st = lib().emu_msk_process_stream(self.h, _p(x), n, n, _p(tags), _p(tag_counts), tcap, _p(syms), _p(err),
                                          _p(mu), _p(bits), cap, _p(prod), _p(cons))
        return dict(syms=syms, bits=bits, err=err, mu=mu, produced=prod, consumed=cons, status=st)

    def general_work(self, noutput, ninput, buf, in_off, tags, nitems_read):
        buf = np.ascontiguousarray(buf, dtype=np.complex64)
        out = np.zeros(max(noutput, 1), np.complex64)
        err = np.zeros(max(noutput, 1), np.float32)
        mu = np.zeros(max(noutput, 1), np.float32)
        bits = np.zeros(max(noutput, 1), np.uint8)
        tags = np.ascontiguousarray(tags, dtype=TAG_DTYPE)
        cons, prod = C.c_int(0), C.c_int(0)
        inp = C.c_void_p(buf.ctypes.data + 8 * in_off)
        st = lib().emu_msk_general_work(self.h, noutput, ninput, inp, _p(out), _p(err), _p(mu), _p(bits), _p(tags),
                                        tags.size, nitems_read, C.byref(cons), C.byref(prod))
        return out[: prod.value], err[: prod.value], mu[: prod.value], bits[: prod.value], cons.value, st


class Agc:
    def __init__(self, nsamples=512, reference=2.0, nchan=1):
        self.h = lib().emu_agc_create(nsamples, reference, nchan)
        self.nchan = nchan

    def __del__(self):
        if getattr(self, "h", None):
            lib().emu_agc_destroy(self.h)
            self.h = None

    def work(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex64).reshape(self.nchan, -1)
        out = np.zeros_like(x)
        lib().emu_agc_process(self.h, _p(x), x.shape[1], _p(out), x.shape[1], x.shape[1])
        return out


class FreqSync:
    def __init__(self, samplerate, bits_per_sec, fftlen=1024, nchan=1, max_items=1 << 20):
        self.h = lib().emu_fs_create(samplerate, bits_per_sec, fftlen, nchan, max_items)
        self.nchan, self.fftlen = nchan, fftlen

    def __del__(self):
        if getattr(self, "h", None):
            lib().emu_fs_destroy(self.h)
            self.h = None

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex64).reshape(self.nchan, -1)
        n = x.shape[1]
        cap = n + self.fftlen
        out = np.zeros((self.nchan, cap), np.complex64)
        nv = cap // self.fftlen + 1
        fh = np.zeros((self.nchan, nv), np.float32)
        m = lib().emu_fs_process(self.h, _p(x), n, n, _p(out), cap, _p(fh), nv)
        return out[:, :m].copy(), fh[:, : m // self.fftlen].copy()

    def freqest_work(self, vecs):
        v = np.ascontiguousarray(vecs, dtype=np.complex64).reshape(self.nchan, -1)
        nvec = v.shape[1] // self.fftlen
        out = np.zeros((self.nchan, max(nvec, 1)), np.float32)
        lib().emu_freqest_work(self.h, _p(v), v.shape[1], _p(out), out.shape[1], nvec)
        return out[:, :nvec]


class Pfb:
    def __init__(self, taps, decim, nstreams=1):
        t = np.ascontiguousarray(taps, dtype=np.float32)
        self.h = lib().emu_pfb_create(decim, _p(t), t.size, nstreams)
        self.decim, self.nstreams = decim, nstreams

    def __del__(self):
        if getattr(self, "h", None):
            lib().emu_pfb_destroy(self.h)
            self.h = None

    def work(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex64).reshape(self.nstreams, -1)
        n = x.shape[1]
        nf = n // self.decim
        out = np.zeros((self.nstreams * 1024, nf), np.complex64)
        lib().emu_pfb_process(self.h, _p(x), n, n, _p(out), nf)
        return out


def fs_agc_process(fs, agc, x):
    """the fused front end under the lane model: (agc output, fhat)"""
    x = np.ascontiguousarray(x, dtype=np.complex64).reshape(fs.nchan, -1)
    n = x.shape[1]
    cap = n + fs.fftlen
    out = np.zeros((fs.nchan, cap), dtype=np.complex64)
    nv = cap // fs.fftlen + 1
    fh = np.zeros((fs.nchan, nv), dtype=np.float32)
    m = lib().emu_fs_agc_process(fs.h, agc.h, _p(x), n, n, _p(out), cap, _p(fh), nv)
    return out[:, :m].copy(), fh[:, : m // fs.fftlen].copy()


def freqest_any(vecs, nchan, fftlen, sample_rate, data_rate):
    """fs_freqest_body for any vector length (the estimator block alone, aisx_freqest_create_n)"""
    v = np.ascontiguousarray(vecs, dtype=np.complex64).reshape(nchan, -1)
    nvec = v.shape[1] // fftlen
    out = np.zeros((nchan, max(nvec, 1)), np.float32)
    lib().emu_freqest_any(_p(v), v.shape[1], _p(out), out.shape[1], nvec, nchan, fftlen, float(sample_rate), int(data_rate))
    return out[:, :nvec]
