"""Host-side mirror of the reference's block interface for the demod hot path.

Same names, constructor arguments and error behaviour as the reference's Python
bindings (swig/ais_swig.i:16-26 exposes ais.corr_est_cc, ais.freqest,
ais.msk_timing_recovery_cc; python/gmsk_sync.py and python/ais_demod.py are
Python hier blocks), with one addition: `nchan`, the number of independent
channels batched through one call.  All arithmetic happens in libaisx.so (HIP,
gfx950); torch is used only for device memory and streams.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import check

TAG_DTYPE = np.dtype([("offset", "<u8"), ("value", "<f8"), ("key", "<i4"), ("chan", "<i4")])


def _stream_ptr(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def _dev_c64(x, nchan):
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.ascontiguousarray(x, dtype=np.complex64))
    if x.dtype != torch.complex64:
        raise TypeError("expected complex64 (gr_complex) items, got %s" % x.dtype)
    if not x.is_cuda:
        x = x.cuda()
    x = x.reshape(nchan, -1)
    if x.stride(1) != 1:
        x = x.contiguous()
    return x


class corr_est_cc:
    """ais.corr_est_cc(symbols, sps, mark_delay, threshold=0.9)
    (include/ais/corr_est_cc.h:102-106, lib/corr_est_cc_impl.cc)."""

    def __init__(self, symbols, sps, mark_delay, threshold=0.9, nchan=1, max_items=65536, max_tags_per_chan=None):
        L = _lib.lib()
        s = np.ascontiguousarray(symbols, dtype=np.complex64)
        self.nchan, self.max_items = int(nchan), int(max_items)
        self._cap = int(max_tags_per_chan) if max_tags_per_chan else max(64, 4 * (self.max_items // 256))
        h = C.c_void_p()
        check(L.aisx_corr_create(C.byref(h), s.ctypes.data_as(C.c_void_p), s.size, float(sps), int(mark_delay),
                                 float(threshold), self.nchan, self.max_items, self._cap), "corr_est_cc")
        self._h = h
        self._N = s.size

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:  # (None while the interpreter shuts down: the process is going anyway)
            _lib.lib().aisx_corr_destroy(h)
            self._h = None

    # -- reference API ---------------------------------------------------
    def symbols(self):
        out = np.zeros(self._N, dtype=np.complex64)
        check(_lib.lib().aisx_corr_symbols(self._h, out.ctypes.data_as(C.c_void_p), out.size), "symbols")
        return out

    def set_symbols(self, symbols):
        s = np.ascontiguousarray(symbols, dtype=np.complex64)
        check(_lib.lib().aisx_corr_set_symbols(self._h, s.ctypes.data_as(C.c_void_p), s.size), "set_symbols")
        self._N = s.size

    def set_lds_claim(self, nbytes):
        """placement knob of the F = 4096 build (include/aisx.h: aisx_corr_set_lds_claim); results do not depend on it"""
        check(_lib.lib().aisx_corr_set_lds_claim(self._h, int(nbytes)), "set_lds_claim")

    def history(self):
        return _lib.lib().aisx_corr_history(self._h)

    def output_multiple(self):
        return _lib.lib().aisx_corr_output_multiple(self._h)

    def max_noutput_items(self):
        return _lib.lib().aisx_corr_max_noutput_items(self._h)

    def threshold(self):
        return _lib.lib().aisx_corr_threshold(self._h)

    def mark_delay(self):
        return _lib.lib().aisx_corr_mark_delay(self._h)

    def nitems_written(self, port=0):
        return _lib.lib().aisx_corr_nitems_written(self._h)

    def reset(self):
        check(_lib.lib().aisx_corr_reset(self._h), "reset")

    # -- batched device path ---------------------------------------------
    def work(self, x, want_corr=False, out=None, stream=None):
        """One work() call on x[nchan][n] new items (device tensor).  Returns
        (out, corr|None); the tags of the call are read with tags()."""
        x = _dev_c64(x, self.nchan)
        n = x.shape[1]
        if out is None:
            out = torch.empty_like(x)
        corr = torch.empty_like(x) if want_corr else None
        check(_lib.lib().aisx_corr_process(self._h, x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0),
                                           corr.data_ptr() if want_corr else None, corr.stride(0) if want_corr else 0,
                                           n, _stream_ptr(stream)), "corr_est_cc.work")
        return out, corr

    def set_profiling(self, on=True):
        check(_lib.lib().aisx_corr_set_profiling(self._h, 1 if on else 0), "set_profiling")

    def last_kernel_ms(self):
        ms = C.c_float(0)
        check(_lib.lib().aisx_corr_last_kernel_ms(self._h, C.byref(ms)), "last_kernel_ms")
        return ms.value

    def kernel_ms_history(self):
        buf = (C.c_float * 64)()
        n = C.c_int(0)
        check(_lib.lib().aisx_corr_kernel_ms_history(self._h, buf, 64, C.byref(n)), "kernel_ms_history")
        return [float(buf[i]) for i in range(n.value)]

    def tags_device(self):
        t, c, cap = C.c_void_p(), C.c_void_p(), C.c_int()
        check(_lib.lib().aisx_corr_tags_device(self._h, C.byref(t), C.byref(c), C.byref(cap)), "tags_device")
        return t, c, cap.value

    def tags(self, stream=None, allow_overflow=False, back=0):
        """Host copy of the last call's tags (back = 1, 2: of the call before / two before):
        structured array (offset,value,key,chan)."""
        cap = self.nchan * self._cap
        buf = np.zeros(cap, dtype=TAG_DTYPE)
        nt = C.c_int(0)
        rc = _lib.lib().aisx_corr_read_tags_back(self._h, int(back), buf.ctypes.data_as(C.c_void_p), cap, C.byref(nt),
                                                 _stream_ptr(stream))
        if not (allow_overflow and rc == _lib.AISX_ERR_OVERFLOW):
            check(rc, "corr_est_cc.tags")
        return buf[: nt.value].copy()

    # -- GNU Radio path (host pointers, one channel) ----------------------
    def work_host(self, in_with_history, noutput_items, nitems_written, want_corr=False, tag_cap=4096):
        a = np.ascontiguousarray(in_with_history, dtype=np.complex64)
        assert a.size >= noutput_items + self._N
        out = np.zeros(noutput_items, dtype=np.complex64)
        corr = np.zeros(noutput_items, dtype=np.complex64) if want_corr else None
        tags = np.zeros(tag_cap, dtype=TAG_DTYPE)
        nt = C.c_int(0)
        check(_lib.lib().aisx_corr_work_host(self._h, a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                             corr.ctypes.data_as(C.c_void_p) if want_corr else None, noutput_items,
                                             nitems_written, tags.ctypes.data_as(C.c_void_p), tag_cap, C.byref(nt)),
              "corr_est_cc.work_host")
        return out, corr, tags[: nt.value].copy()


class msk_timing_recovery_cc:
    """ais.msk_timing_recovery_cc(sps, gain, limit, osps)
    (include/ais/msk_timing_recovery_cc.h:60-69, lib/msk_timing_recovery_cc_impl.cc)."""

    def __init__(self, sps, gain, limit, osps=1, nchan=1, max_items=65536):
        L = _lib.lib()
        self.nchan, self.max_items = int(nchan), int(max_items)
        h = C.c_void_p()
        check(L.aisx_msk_create(C.byref(h), float(sps), float(gain), float(limit), int(osps), self.nchan,
                                self.max_items), "msk_timing_recovery_cc")
        self._h = h
        self.out_capacity = L.aisx_msk_out_capacity(h)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:  # (None while the interpreter shuts down: the process is going anyway)
            _lib.lib().aisx_msk_destroy(h)
            self._h = None

    def set_gain(self, gain):
        check(_lib.lib().aisx_msk_set_gain(self._h, float(gain)), "set_gain")

    def get_gain(self):
        return _lib.lib().aisx_msk_get_gain(self._h)

    def set_limit(self, limit):
        check(_lib.lib().aisx_msk_set_limit(self._h, float(limit)), "set_limit")
        self.out_capacity = _lib.lib().aisx_msk_out_capacity(self._h)

    def get_limit(self):
        return _lib.lib().aisx_msk_get_limit(self._h)

    def set_sps(self, sps):
        check(_lib.lib().aisx_msk_set_sps(self._h, float(sps)), "set_sps")
        self.out_capacity = _lib.lib().aisx_msk_out_capacity(self._h)

    def get_sps(self):
        return _lib.lib().aisx_msk_get_sps(self._h)

    def forecast(self, noutput_items):
        return _lib.lib().aisx_msk_forecast(self._h, int(noutput_items))

    def reset(self):
        check(_lib.lib().aisx_msk_reset(self._h), "reset")

    def work(self, x, tags_from=None, want_syms=True, want_aux=False, want_bits=True, stream=None, outs=None,
             tags_ptrs=None):
        """One stream step on x[nchan][n] new items.  `tags_from` = the
        corr_est_cc block whose last call produced the time_est tags (or None).
        Returns dict(syms, err, mu, bits, produced) of device tensors."""
        x = _dev_c64(x, self.nchan)
        n = x.shape[1]
        cap = self.out_capacity
        dev = x.device
        o = outs or {}
        syms = o.get("syms") if "syms" in o else (torch.empty((self.nchan, cap), dtype=torch.complex64, device=dev) if want_syms else None)
        err = o.get("err") if "err" in o else (torch.empty((self.nchan, cap), dtype=torch.float32, device=dev) if want_aux else None)
        mu = o.get("mu") if "mu" in o else (torch.empty((self.nchan, cap), dtype=torch.float32, device=dev) if want_aux else None)
        bits = o.get("bits") if "bits" in o else (torch.empty((self.nchan, cap), dtype=torch.uint8, device=dev) if want_bits else None)
        prod = o.get("produced") if "produced" in o else torch.empty(self.nchan, dtype=torch.int32, device=dev)
        stride = None
        for name, t, dt in (("syms", syms, torch.complex64), ("err", err, torch.float32), ("mu", mu, torch.float32),
                            ("bits", bits, torch.uint8)):
            if t is None:
                continue
            if (t.dim() != 2 or t.shape[0] != self.nchan or t.dtype != dt or t.stride(1) != 1 or not t.is_cuda
                    or t.shape[1] < 1 or t.stride(0) < t.shape[1]):
                raise ValueError("msk_timing_recovery_cc.work: outs[%r] must be a (nchan, cap) %s device tensor with "
                                 "unit inner stride" % (name, dt))
            if stride is None:
                stride, width = t.stride(0), t.shape[1]
            elif t.stride(0) != stride:
                raise ValueError("msk_timing_recovery_cc.work: all output tensors must share one row stride")
            else:
                width = min(width, t.shape[1])
        if prod.dim() != 1 or prod.shape[0] < self.nchan or prod.dtype != torch.int32 or not prod.is_cuda:
            raise ValueError("msk_timing_recovery_cc.work: outs['produced'] must be an int32 device tensor of nchan items")
        if stride is None:
            stride = width = cap
        if width < stride:
            # rows narrower than their stride (a column slice): the kernel bounds its writes by the
            # stride it is given, so hand it rows it owns entirely
            raise ValueError("msk_timing_recovery_cc.work: output rows must be as wide as their stride")
        if tags_ptrs is not None:
            tptr, cptr, tcap = tags_ptrs  # as returned by corr_est_cc.tags_device() right after its work()
        elif tags_from is not None:
            tptr, cptr, tcap = tags_from.tags_device()
        else:
            tptr, cptr, tcap = None, None, 0
        check(_lib.lib().aisx_msk_process_stream(
            self._h, x.data_ptr(), x.stride(0), n, tptr, cptr, tcap,
            syms.data_ptr() if syms is not None else None, err.data_ptr() if err is not None else None,
            mu.data_ptr() if mu is not None else None, bits.data_ptr() if bits is not None else None, stride,
            prod.data_ptr(), _stream_ptr(stream)), "msk_timing_recovery_cc.work")
        return dict(syms=syms, err=err, mu=mu, bits=bits, produced=prod)

    def set_tail_stream(self, stream):
        """Run the NRZI bit tail of every work() call on `stream` (None: back on the call's own
        stream).  The caller then alternates between two sets of output buffers (outs=) and orders
        its readers of `bits` after that stream (wait_tail)."""
        check(_lib.lib().aisx_msk_set_tail_stream(self._h, _stream_ptr(stream) if stream is not None else None,
                                                  1 if stream is not None else 0), "set_tail_stream")

    def wait_tail(self, stream=None):
        check(_lib.lib().aisx_msk_wait_tail(self._h, _stream_ptr(stream)), "wait_tail")

    def wait_prepass(self, stream=None):
        """`stream` waits until the last work() call's recovery kernel stands at the head of its queue
        (aisx_msk_wait_prepass): call it on the stream whose next kernels would otherwise take the
        LDS the recovery's workgroups need.  The first call only arms the event."""
        check(_lib.lib().aisx_msk_wait_prepass(self._h, _stream_ptr(stream)), "wait_prepass")

    def last_status(self, stream=None):
        st = C.c_int(0)
        check(_lib.lib().aisx_msk_last_status(self._h, C.byref(st), _stream_ptr(stream)), "last_status")
        return st.value

    def set_time_parallel(self, restart_points=64, join_kernel=-1, max_unit_items=0):
        """The time-parallel recovery (include/aisx.h: aisx_msk_set_time_parallel); restart_points=0 switches it off."""
        check(_lib.lib().aisx_msk_set_time_parallel(self._h, int(restart_points), int(join_kernel), int(max_unit_items)),
              "set_time_parallel")

    def set_max_noutput_items(self, m):
        """gr::block::set_max_noutput_items(): output items one general_work call is offered at most (0: what fits)."""
        check(_lib.lib().aisx_msk_set_max_noutput_items(self._h, int(m)), "set_max_noutput_items")

    def max_noutput_items(self):
        return _lib.lib().aisx_msk_get_max_noutput_items(self._h)

    def set_profiling(self, on=True):
        check(_lib.lib().aisx_msk_set_profiling(self._h, 1 if on else 0), "set_profiling")

    def kernel_ms_history(self):
        buf = (C.c_float * 64)()
        n = C.c_int(0)
        check(_lib.lib().aisx_msk_kernel_ms_history(self._h, buf, 64, C.byref(n)), "kernel_ms_history")
        return [float(buf[i]) for i in range(n.value)]

    def restart_stats(self, stream=None):
        """What the time-parallel recovery made of the last call (sums over the channels)."""
        a = (C.c_longlong * 10)()
        check(_lib.lib().aisx_msk_restart_stats(self._h, a, _stream_ptr(stream)), "restart_stats")
        return dict(restart_points=a[0], units_taken=a[1], symbols_from_units=a[2], units_ended_at_next=a[3],
                    units_ended_elsewhere=a[4], calls=a[5], links_equal=a[6], links=a[7], longest_unit_items=a[8],
                    unit_items=a[9])

    def general_work_host(self, noutput_items, ninput_items, buf, in_off, tags, nitems_read, in_has_lookahead=True):
        """GNU Radio path: in = &buf[in_off]; tags: structured array (TAG_DTYPE)."""
        buf = np.ascontiguousarray(buf, dtype=np.complex64)
        out = np.zeros(max(noutput_items, 1), np.complex64)
        err = np.zeros(max(noutput_items, 1), np.float32)
        mu = np.zeros(max(noutput_items, 1), np.float32)
        bits = np.zeros(max(noutput_items, 1), np.uint8)
        tags = np.ascontiguousarray(tags, dtype=TAG_DTYPE)
        cons, prod = C.c_int(0), C.c_int(0)
        inp = C.c_void_p(buf.ctypes.data + 8 * in_off)
        check(_lib.lib().aisx_msk_general_work_host(
            self._h, noutput_items, ninput_items, inp, out.ctypes.data_as(C.c_void_p), err.ctypes.data_as(C.c_void_p),
            mu.ctypes.data_as(C.c_void_p), bits.ctypes.data_as(C.c_void_p), tags.ctypes.data_as(C.c_void_p), tags.size,
            nitems_read, 1 if in_has_lookahead else 0, C.byref(cons), C.byref(prod)), "general_work_host")
        p = prod.value
        return out[:p], err[:p], mu[:p], bits[:p], cons.value


class square_and_fft_sync_cc:
    """ais.square_and_fft_sync_cc(samplerate, bits_per_sec, fftlen) (python/gmsk_sync.py:14-37),
    which owns ais.freqest(int(samplerate), int(bits_per_sec), fftlen) (lib/freqest_impl.cc)."""

    def __init__(self, samplerate, bits_per_sec, fftlen, nchan=1, max_items=65536):
        self.nchan, self.fftlen, self.max_items = int(nchan), int(fftlen), int(max_items)
        h = C.c_void_p()
        check(_lib.lib().aisx_freqsync_create(C.byref(h), float(samplerate), float(bits_per_sec), self.fftlen,
                                              self.nchan, self.max_items), "square_and_fft_sync_cc")
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:  # (None while the interpreter shuts down: the process is going anyway)
            _lib.lib().aisx_freqsync_destroy(h)
            self._h = None

    def reset(self):
        check(_lib.lib().aisx_freqsync_reset(self._h), "reset")

    def estimate_ahead(self, x, stream=None, walk_stream=None):
        """Prepare the frequency estimates (on `stream`) and the NCO phase walk (on `walk_stream`,
        default: the same) of the next freq_sync_agc() call on exactly this tensor
        (aisx_freqsync_estimate_ahead): the serial walk of step k + 1 then runs beside the sample
        passes of step k.  Up to two preparations may wait (the second only behind a call whose
        length is a multiple of fftlen with nothing pending); they are consumed in order."""
        x = _dev_c64(x, self.nchan)
        check(_lib.lib().aisx_freqsync_estimate_ahead(self._h, x.data_ptr(), x.stride(0), x.shape[1], _stream_ptr(stream),
                                                      _stream_ptr(walk_stream) if walk_stream is not None else None),
              "estimate_ahead")

    def work(self, x, want_fhat=False, stream=None, out=None):
        """`out`: optional preallocated (nchan, >= n + fftlen) complex64 buffer the result is a view of."""
        x = _dev_c64(x, self.nchan)
        n = x.shape[1]
        cap = n + self.fftlen
        if out is None:
            out = torch.empty((self.nchan, cap), dtype=torch.complex64, device=x.device)
        elif out.shape[0] != self.nchan or out.shape[1] < cap or out.dtype != torch.complex64 or out.stride(1) != 1:
            raise ValueError("out must be a (nchan, >= n + fftlen) complex64 buffer")
        nv = cap // self.fftlen + 1
        fh = torch.empty((self.nchan, nv), dtype=torch.float32, device=x.device) if want_fhat else None
        nout = C.c_int(0)
        check(_lib.lib().aisx_freqsync_process(self._h, x.data_ptr(), x.stride(0), n, out.data_ptr(), out.stride(0),
                                               fh.data_ptr() if want_fhat else None, nv if want_fhat else 0,
                                               C.byref(nout), _stream_ptr(stream)), "square_and_fft_sync_cc.work")
        m = nout.value
        return out[:, :m], (fh[:, : m // self.fftlen] if want_fhat else None)


def freq_sync_agc(freq_sync, agc, x, want_fhat=False, stream=None, out=None):
    """freq_sync -> agc (the first two blocks of python/ais_demod.py:56) in one pass: the same
    result, bit for bit, as agc.work(freq_sync.work(x)[0]), without the hier block's output being
    stored (aisx_freqsync_agc_process).  Returns (agc output, fhat | None)."""
    x = _dev_c64(x, freq_sync.nchan)
    n = x.shape[1]
    cap = n + freq_sync.fftlen
    if out is None:
        out = torch.empty((freq_sync.nchan, cap), dtype=torch.complex64, device=x.device)
    elif out.shape[0] != freq_sync.nchan or out.shape[1] < cap or out.dtype != torch.complex64 or out.stride(1) != 1:
        raise ValueError("out must be a (nchan, >= n + fftlen) complex64 buffer")
    nv = cap // freq_sync.fftlen + 1
    fh = torch.empty((freq_sync.nchan, nv), dtype=torch.float32, device=x.device) if want_fhat else None
    nout = C.c_int(0)
    check(_lib.lib().aisx_freqsync_agc_process(freq_sync._h, agc._h, x.data_ptr(), x.stride(0), n, out.data_ptr(),
                                               out.stride(0), fh.data_ptr() if want_fhat else None, nv if want_fhat else 0,
                                               C.byref(nout), _stream_ptr(stream)), "freq_sync_agc")
    m = nout.value
    return out[:, :m], (fh[:, : m // freq_sync.fftlen] if want_fhat else None)


class freqest:
    """ais.freqest(sample_rate, data_rate, fftlen) (include/ais/freqest.h:49): consumes
    fft-shifted spectra, one float estimate per vector."""

    def __init__(self, sample_rate, data_rate, fftlen, nchan=1):
        # the block's own make(): d_offset / d_binsize from the float rate (lib/freqest_impl.cc:46-47), any fftlen >= 2
        self.nchan, self.fftlen = int(nchan), int(fftlen)
        h = C.c_void_p()
        check(_lib.lib().aisx_freqest_create_n(C.byref(h), float(sample_rate), int(data_rate), self.fftlen, self.nchan, 64), "freqest")
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.lib().aisx_freqsync_destroy(self._h)
            self._h = None

    def work(self, vecs, stream=None):
        v = _dev_c64(vecs, self.nchan)
        nvec = v.shape[1] // self.fftlen
        out = torch.empty((self.nchan, max(nvec, 1)), dtype=torch.float32, device=v.device)
        check(_lib.lib().aisx_freqest_work(self._h, v.data_ptr(), v.stride(0), out.data_ptr(), out.stride(0), nvec,
                                           _stream_ptr(stream)), "freqest.work")
        return out[:, :nvec]

    def work_host(self, vecs):
        """GNU Radio path: `vecs` = input_items[0], whole fftlen-vectors on the host; returns
        output_items[0] (one float per vector).  One call = one freqest::work call."""
        v = np.ascontiguousarray(vecs, dtype=np.complex64).reshape(-1)
        nvec = v.size // self.fftlen
        out = np.zeros(max(nvec, 1), dtype=np.float32)
        got = check(_lib.lib().aisx_freqest_work_host(self._h, nvec, v.ctypes.data_as(C.c_void_p),
                                                      out.ctypes.data_as(C.c_void_p)), "freqest.work_host")
        return out[:got]


class feedforward_agc_cc:
    """analog.feedforward_agc_cc(nsamples, reference) as used at python/ais_demod.py:35."""

    def __init__(self, nsamples, reference, nchan=1, max_items=65536):
        self.nchan = int(nchan)
        h = C.c_void_p()
        check(_lib.lib().aisx_agc_create(C.byref(h), int(nsamples), float(reference), self.nchan, int(max_items)),
              "feedforward_agc_cc")
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:  # (None while the interpreter shuts down: the process is going anyway)
            _lib.lib().aisx_agc_destroy(h)
            self._h = None

    def reset(self):
        check(_lib.lib().aisx_agc_reset(self._h), "reset")

    def set_floor(self, floor_env):
        """initial max_env of the window search: 1e-4 (GNU Radio 3.7/3.8, the default) or 1e-12."""
        check(_lib.lib().aisx_agc_set_floor(self._h, float(floor_env)), "set_floor")

    def set_streaming(self, on):
        """on = False: the tile kernels for every call (the streaming kernel serves the stock window otherwise)."""
        check(_lib.lib().aisx_agc_set_streaming(self._h, 1 if on else 0), "set_streaming")

    def work(self, x, out=None, stream=None):
        x = _dev_c64(x, self.nchan)
        if out is None:
            out = torch.empty_like(x)
        check(_lib.lib().aisx_agc_process(self._h, x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0),
                                          x.shape[1], _stream_ptr(stream)), "feedforward_agc_cc.work")
        return out


_hwq_warned = False


def _warn_if_few_hw_queues():
    """The chain keeps four streams busy; the HIP runtime reads GPU_MAX_HW_QUEUES (default 4, shared by every
    stream of the process) at its first call.  Correct results either way -- but the streams then run in turn."""
    global _hwq_warned
    import os
    import warnings

    from . import HW_QUEUES_SET_TOO_LATE

    try:
        few = int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8
    except ValueError:
        few = True
    if (HW_QUEUES_SET_TOO_LATE or few) and not _hwq_warned:
        _hwq_warned = True
        warnings.warn("ais_amd: the pipelined chain wants GPU_MAX_HW_QUEUES >= 8 in the environment BEFORE the first HIP call "
                      "(%s); with the runtime's four hardware queues its streams share queues and the step is slower"
                      % ("the process had initialised torch.cuda before ais_amd was imported" if HW_QUEUES_SET_TOO_LATE
                         else "it is %s" % os.environ.get("GPU_MAX_HW_QUEUES", "unset")), RuntimeWarning, stacklevel=3)


class ais_demod:
    """ais.ais_demod(options) (python/ais_demod.py:21-56): the demod chain
    freq_sync -> agc -> corr_est (preamble_detect) -> msk timing recovery (clockrec)
    -> quadrature demod -> slicer -> diff decoder -> invert (:56), with the same
    option keys and constants, for `nchan` channels at once.

    `stages` selects the upstream conditioning: 'stock' (freq_sync + agc, the
    reference's connect order) or 'core' (corr_est -> msk only, the chain
    BASELINE.json's metric names)."""

    def __init__(self, options, nchan=1, max_items=65536, stages="stock", preamble_symbols=None, fused_front_end=False):
        from .modulate import gmsk_mod, modulate_vector_bc

        self._samples_per_symbol = options["samples_per_symbol"]
        self._bits_per_sec = options["bits_per_sec"]
        self._samplerate = self._samples_per_symbol * self._bits_per_sec
        self._clockrec_gain = options["clockrec_gain"]
        self._omega_relative_limit = options["omega_relative_limit"]
        self.fftlen = options["fftlen"]
        self.nchan = nchan
        self._max_items = int(max_items)
        self._chain = None
        self.stages = stages
        # freq_sync -> agc in one pass (freq_sync_agc): same results, bit for bit; off by default: the
        # separate NCO phase walk it needs is slower than fs_mix's in-kernel one when everything runs
        # in series on one stream -- it pays when the caller prepares it ahead (estimate_ahead, bench.py)
        self.fused_front_end = bool(fused_front_end)
        if stages == "stock":
            self.freq_sync = square_and_fft_sync_cc(self._samplerate, self._bits_per_sec, self.fftlen, nchan=nchan,
                                                    max_items=max_items)
            self.agc = feedforward_agc_cc(512, 2, nchan=nchan, max_items=max_items + self.fftlen)
        else:
            self.freq_sync = self.agc = None
        if preamble_symbols is None:
            self.preamble = [1, 1, 0, 0] * 7
            self.mod = gmsk_mod(int(self._samples_per_symbol), 0.4)
            self.mod_vector = modulate_vector_bc(self.mod.to_basic_block(), self.preamble, [1])
        else:
            self.mod_vector = np.asarray(preamble_symbols, dtype=np.complex64)
        self.preamble_detect = corr_est_cc(self.mod_vector, self._samples_per_symbol, 1, 0.9, nchan=nchan,
                                           max_items=max_items + self.fftlen)
        self.clockrec = msk_timing_recovery_cc(self._samples_per_symbol, self._clockrec_gain,
                                               self._omega_relative_limit, 1, nchan=nchan,
                                               max_items=max_items + self.fftlen)

    # -- pipelined step (aisx_chain_*): the path bench.py times ------------------------------
    def _chain_handle(self):
        if getattr(self, "_chain", None) is None:
            _warn_if_few_hw_queues()
            h = C.c_void_p()
            stock = self.stages == "stock"
            check(_lib.lib().aisx_chain_create(C.byref(h), self.freq_sync._h if stock else None, self.agc._h if stock else None,
                                               self.preamble_detect._h, self.clockrec._h, self.nchan, self._max_items,
                                               self.fftlen if stock else 0), "ais_demod chain")
            self._chain = h
            self._chain_outs = []
            self._chain_step = -1
        return self._chain

    def __del__(self):
        h = getattr(self, "_chain", None)
        if h and _lib is not None:
            _lib.lib().aisx_chain_destroy(h)  # (before the stage handles it borrows go)
            self._chain = None

    def work_pipelined(self, x, x_next=None, want_syms=False, outs=None, stream=None):
        """One step of the pipelined chain (aisx_chain_step) on x[nchan][n]: the same results as work(), bit
        for bit, with the sample passes of this step running beside the timing recovery of the
        previous one on the chain's own streams.  `x_next` = the NEXT step's input if it is
        already on the device (its frequency estimates and NCO phase walk are then prepared
        during this step; the next call must pass exactly that tensor, unchanged).
        Returns dict(bits, produced[, syms], step): device tensors that are complete once
        wait(step) has returned; AISX_CHAIN_DEPTH sets rotate, so a result must be consumed
        before the call three steps later.  Do not mix with work() on one object."""
        h = self._chain_handle()
        x = _dev_c64(x, self.nchan)
        nx = _dev_c64(x_next, self.nchan) if x_next is not None else None
        # (the chain reads its inputs asynchronously, on streams torch's allocator does not know of:
        # the tensors of the steps that may still be in flight stay referenced here)
        keep = getattr(self, "_chain_keep", None) or []
        keep.append((x, nx))
        self._chain_keep = keep[-(_lib.lib().aisx_chain_depth() + 1):]
        cap = self.clockrec.out_capacity
        if outs is None:
            depth = _lib.lib().aisx_chain_depth()
            while len(self._chain_outs) < depth:
                self._chain_outs.append(dict(
                    syms=torch.empty((self.nchan, cap), dtype=torch.complex64, device=x.device) if want_syms else None,
                    bits=torch.empty((self.nchan, cap), dtype=torch.uint8, device=x.device),
                    produced=torch.empty(self.nchan, dtype=torch.int32, device=x.device)))
            outs = self._chain_outs[(self._chain_step + 1) % depth]
            if want_syms and outs["syms"] is None:
                outs["syms"] = torch.empty((self.nchan, cap), dtype=torch.complex64, device=x.device)
        syms, bits, prod = outs.get("syms") if want_syms else None, outs["bits"], outs["produced"]
        step = C.c_longlong(0)
        check(_lib.lib().aisx_chain_step(
            h, x.data_ptr(), x.stride(0), x.shape[1], nx.data_ptr() if nx is not None else None,
            nx.stride(0) if nx is not None else 0, nx.shape[1] if nx is not None else 0,
            syms.data_ptr() if syms is not None else None, bits.data_ptr(), bits.stride(0), prod.data_ptr(),
            _stream_ptr(stream), C.byref(step)), "ais_demod.work_pipelined")
        self._chain_step = step.value
        return dict(bits=bits, produced=prod, syms=syms, step=step.value)

    def wait(self, step=None, stream=None, host=False):
        """`stream` (default: the current one) -- or the calling thread with host=True -- waits for the
        outputs of `step` (default: the last one issued)."""
        check(_lib.lib().aisx_chain_wait(self._chain_handle(), self._chain_step if step is None else step,
                                         None if host else _stream_ptr(stream), 1 if host else 0), "ais_demod.wait")

    def synchronize(self):
        check(_lib.lib().aisx_chain_synchronize(self._chain_handle()), "ais_demod.synchronize")

    def corr_output(self, step, chan0=0, nch=None, stream=None):
        """corr_est's delayed output of `step` (one of the last AISX_CHAIN_DEPTH), rows chan0 .. chan0 + nch - 1"""
        nch = self.nchan - chan0 if nch is None else nch
        out = torch.empty((nch, self._max_items + self.fftlen), dtype=torch.complex64, device="cuda")
        n = C.c_int(0)
        check(_lib.lib().aisx_chain_read_corr_output(self._chain_handle(), step, chan0, nch, out.data_ptr(), out.stride(0),
                                                     C.byref(n), _stream_ptr(stream)), "ais_demod.corr_output")
        return out[:, : n.value]

    def step_tags(self, step, stream=None):
        """corr_est's tags of pipelined step `step` (one of the last AISX_CHAIN_DEPTH) on the host, by step
        number (aisx_chain_read_tags): a step whose front end emitted no whole vector has none."""
        cap = self.nchan * self.preamble_detect._cap
        buf = np.zeros(cap, dtype=TAG_DTYPE)
        nt = C.c_int(0)
        check(_lib.lib().aisx_chain_read_tags(self._chain_handle(), int(step), buf.ctypes.data_as(C.c_void_p), cap, C.byref(nt),
                                              _stream_ptr(stream)), "ais_demod.step_tags")
        return buf[: nt.value].copy()

    def chain_stream(self, which):
        """the chain's streams as raw hipStream_t values: 0 sample passes, 1 timing recovery, 2 bit tail, 3 phase walk"""
        return _lib.lib().aisx_chain_stream(self._chain_handle(), which)

    def work(self, x, want_syms=False, stream=None):
        """One chain step on x[nchan][n].  Returns dict(bits, produced[, syms])."""
        y = _dev_c64(x, self.nchan)
        if self.stages == "stock":
            if self.fused_front_end:
                y, _ = freq_sync_agc(self.freq_sync, self.agc, y, stream=stream)
            else:
                y, _ = self.freq_sync.work(y, stream=stream)
            if y.shape[1] == 0:
                return dict(bits=None, produced=None, syms=None)
            if not self.fused_front_end:
                y = self.agc.work(y, stream=stream)
        y, _ = self.preamble_detect.work(y, stream=stream)
        r = self.clockrec.work(y, tags_from=self.preamble_detect, want_syms=want_syms, stream=stream)
        return r


def firdes_low_pass(gain, sampling_freq, cutoff_freq, transition_width):
    """filter.firdes.low_pass(gain, fs, cutoff, transition) with the default Hamming
    window, as python/radio.py:51 calls it ([GR] firdes.cc: windowed sinc, DC gain
    normalised).  Host side, init only."""
    ntaps = int(53.0 * sampling_freq / (22.0 * transition_width))
    if (ntaps & 1) == 0:
        ntaps += 1
    m = (ntaps - 1) // 2
    n = np.arange(-m, m + 1, dtype=np.float64)
    w = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(ntaps) / (ntaps - 1))
    fwt0 = 2 * np.pi * cutoff_freq / sampling_freq
    with np.errstate(invalid="ignore", divide="ignore"):
        taps = np.where(n == 0, fwt0 / np.pi, np.sin(n * fwt0) / (n * np.pi)) * w
    taps = taps.astype(np.float32).astype(np.float64)
    fmax = taps[m] + 2 * taps[m + 1:].sum()
    return (taps * (gain / fmax)).astype(np.float32)


class pfb_channelizer_ccf:
    """Wideband front end (BASELINE config 5): all `nlanes` uniformly spaced channels of
    one wideband stream at once; lane m is what
    freq_xlating_fir_filter_ccf(decim, taps, m*fs/nlanes, fs) (python/radio.py:52-54)
    would produce."""

    def __init__(self, nlanes, taps, decim=None, nstreams=1, max_frames=4096):
        t = np.ascontiguousarray(taps, dtype=np.float32)
        self.nlanes, self.decim, self.nstreams = int(nlanes), int(decim or nlanes), int(nstreams)
        h = C.c_void_p()
        check(_lib.lib().aisx_pfb_create(C.byref(h), self.nlanes, self.decim, t.ctypes.data_as(C.c_void_p), t.size,
                                         self.nstreams, int(max_frames)), "pfb_channelizer_ccf")
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:  # (None while the interpreter shuts down: the process is going anyway)
            _lib.lib().aisx_pfb_destroy(h)
            self._h = None

    def work(self, x, stream=None):
        x = _dev_c64(x, self.nstreams)
        n = x.shape[1]
        nf = n // self.decim
        out = torch.empty((self.nstreams * self.nlanes, max(nf, 1)), dtype=torch.complex64, device=x.device)
        got = C.c_int(0)
        check(_lib.lib().aisx_pfb_process(self._h, x.data_ptr(), x.stride(0), n, out.data_ptr(), out.stride(0),
                                          C.byref(got), _stream_ptr(stream)), "pfb_channelizer_ccf.work")
        return out[:, : got.value]
