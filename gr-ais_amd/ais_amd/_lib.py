"""ctypes loader for libaisx.so (the HIP kernels + C ABI, include/aisx.h).

There is deliberately no fallback: if the shared library is missing the import
fails, and if no MI355X is visible every block constructor raises.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# The product library reads no environment variable.  lib/libaisx_exp.so is the same code built with -DAISX_EXPERIMENTS
# (alternative kernels, AISX_* knobs): a process asks for it BEFORE the first call with AISX_LIB_VARIANT=exp (the twin tests
# under tests/exp_builds/, tools/ab_bench.py) -- a choice of this loader, not of the library.
LIB_VARIANT = os.environ.get("AISX_LIB_VARIANT", "")
LIB_PATH = os.path.join(os.path.dirname(_PKG), "lib", "libaisx_exp.so" if LIB_VARIANT == "exp" else "libaisx.so")

AISX_OK = 0
AISX_ERR_INVALID = -1
AISX_ERR_OUT_OF_RANGE = -2
AISX_ERR_HIP = -3
AISX_ERR_NO_DEVICE = -4
AISX_ERR_OVERFLOW = -5
AISX_ERR_RUNTIME = -6

KEY_CORR_START, KEY_PHASE_EST, KEY_TIME_EST, KEY_CORR_EST, KEY_PORT1 = 0, 1, 2, 3, 0x100
KEY_NAMES = {0: "corr_start", 1: "phase_est", 2: "time_est", 3: "corr_est"}


class AisxError(RuntimeError):
    pass


class NoDeviceError(AisxError):
    pass


_lib = None
_torch_first = False


def lib(device=True):
    """libaisx.so, loaded once.  device=False: a host-only helper is asking (HDLC deframer, NMEA: plain C++
    in the same library) -- torch is then not imported on its account."""
    global _lib, _torch_first
    if _lib is not None:
        if device and not _torch_first:
            import sys
            import warnings

            if "torch" not in sys.modules:
                warnings.warn("ais_amd: libaisx.so was loaded by a host-only helper before torch; torch ships its own copy of "
                              "the HIP runtime, and a process with two of them may see no device through the second: import "
                              "torch (or ais_amd.blocks) first in processes that use both", RuntimeWarning, stacklevel=2)
            _torch_first = True  # (say it once)
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libaisx.so not built (%s): run `make -C gr-ais_amd` or __graft_entry__.build(); "
            "there is no CPU fallback" % LIB_PATH)
    # torch ships its own copy of the HIP runtime: load it FIRST, so that libaisx.so binds to the
    # runtime that owns the devices torch hands out pointers of (the other order leaves this
    # process with two runtimes, and the second one sees no device)
    import sys

    if device or "torch" in sys.modules:
        try:
            import torch  # noqa: F401

            _torch_first = True
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, f32, f64, u64, lng = C.c_void_p, C.c_int, C.c_uint, C.c_float, C.c_double, C.c_uint64, C.c_long
    pvp, pi32 = C.POINTER(C.c_void_p), C.POINTER(C.c_int)

    def sig(name, res, args):
        f = getattr(L, name)  # AttributeError = libaisx.so is older than this binding: rebuild
        f.restype = res
        f.argtypes = args

    sig("aisx_version", i32, [])
    sig("aisx_last_error", C.c_char_p, [])
    sig("aisx_device_count", i32, [pi32])
    sig("aisx_set_device", i32, [i32])
    sig("aisx_util_copy_GBs", i32, [C.c_size_t, i32, C.POINTER(C.c_float)])
    sig("aisx_util_agc_rcp_mismatches", i32, [f32, C.POINTER(C.c_ulonglong), C.POINTER(C.c_float)])
    sig("aisx_corr_create", i32, [pvp, vp, i32, f32, u32, f32, i32, i32, i32])
    sig("aisx_corr_destroy", i32, [vp])
    sig("aisx_corr_symbols", i32, [vp, vp, i32])
    sig("aisx_corr_set_symbols", i32, [vp, vp, i32])
    sig("aisx_corr_geometry", i32, [vp, pi32, pi32])
    sig("aisx_corr_history", i32, [vp])
    sig("aisx_corr_output_multiple", i32, [vp])
    sig("aisx_corr_max_noutput_items", i32, [vp])
    sig("aisx_corr_threshold", f32, [vp])
    sig("aisx_corr_mark_delay", u32, [vp])
    sig("aisx_corr_nitems_written", u64, [vp])
    sig("aisx_corr_reset", i32, [vp])
    sig("aisx_corr_process", i32, [vp, vp, lng, vp, lng, vp, lng, i32, vp])
    sig("aisx_corr_set_lds_claim", i32, [vp, i32])
    sig("aisx_corr_set_profiling", i32, [vp, i32])
    sig("aisx_corr_last_kernel_ms", i32, [vp, C.POINTER(C.c_float)])
    sig("aisx_corr_kernel_ms_history", i32, [vp, C.POINTER(C.c_float), i32, pi32])
    sig("aisx_corr_tags_device", i32, [vp, pvp, pvp, pi32])
    sig("aisx_corr_read_tags", i32, [vp, vp, i32, pi32, vp])
    sig("aisx_corr_read_tags_back", i32, [vp, i32, vp, i32, pi32, vp])
    sig("aisx_corr_work_host", i32, [vp, vp, vp, vp, i32, u64, vp, i32, pi32])
    sig("aisx_msk_create", i32, [pvp, f32, f32, f32, i32, i32, i32])
    sig("aisx_msk_destroy", i32, [vp])
    sig("aisx_msk_set_gain", i32, [vp, f32])
    sig("aisx_msk_get_gain", f32, [vp])
    sig("aisx_msk_set_limit", i32, [vp, f32])
    sig("aisx_msk_get_limit", f32, [vp])
    sig("aisx_msk_set_sps", i32, [vp, f32])
    sig("aisx_msk_get_sps", f32, [vp])
    sig("aisx_msk_forecast", i32, [vp, i32])
    sig("aisx_msk_out_capacity", i32, [vp])
    sig("aisx_msk_reset", i32, [vp])
    sig("aisx_msk_process_stream", i32, [vp, vp, lng, i32, vp, vp, i32, vp, vp, vp, vp, lng, vp, vp])
    sig("aisx_msk_process_stream_after", i32, [vp, vp, lng, i32, vp, vp, i32, vp, vp, vp, vp, lng, vp, vp, vp])
    sig("aisx_msk_last_status", i32, [vp, pi32, vp])
    sig("aisx_msk_restart_stats", i32, [vp, vp, vp])
    sig("aisx_msk_set_profiling", i32, [vp, i32])
    sig("aisx_msk_kernel_ms_history", i32, [vp, C.POINTER(C.c_float), i32, pi32])
    sig("aisx_msk_set_max_noutput_items", i32, [vp, i32])
    sig("aisx_msk_set_time_parallel", i32, [vp, i32, i32, i32])
    sig("aisx_msk_get_max_noutput_items", i32, [vp])
    sig("aisx_msk_set_tail_stream", i32, [vp, vp, i32])
    sig("aisx_msk_wait_tail", i32, [vp, vp])
    sig("aisx_msk_wait_prepass", i32, [vp, vp])
    sig("aisx_msk_set_head_start", i32, [vp, i32])
    sig("aisx_msk_geometry", i32, [vp, pi32, pi32])
    sig("aisx_msk_general_work_host", i32, [vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, u64, i32, pi32, pi32])
    sig("aisx_freqsync_create", i32, [pvp, f64, f64, i32, i32, i32])
    sig("aisx_freqest_create", i32, [pvp, f32, i32, i32, i32])
    sig("aisx_freqest_create_n", i32, [pvp, f32, i32, i32, i32, i32])
    sig("aisx_freqsync_is_estimator_only", i32, [vp])
    sig("aisx_freqsync_geometry", i32, [vp, pi32, pi32, pi32])
    sig("aisx_freqsync_drop_ahead", i32, [vp, vp])
    sig("aisx_freqsync_destroy", i32, [vp])
    sig("aisx_freqsync_reset", i32, [vp])
    sig("aisx_freqsync_process", i32, [vp, vp, lng, i32, vp, lng, vp, lng, pi32, vp])
    sig("aisx_freqest_work", i32, [vp, vp, lng, vp, lng, i32, vp])
    sig("aisx_freqest_work_host", i32, [vp, i32, vp, vp])
    sig("aisx_freqsync_work_host", i32, [vp, vp, i32, vp, i32, vp, i32])
    sig("aisx_agc_work_host", i32, [vp, i32, vp, vp])
    sig("aisx_freqsync_agc_process", i32, [vp, vp, vp, lng, i32, vp, lng, vp, lng, pi32, vp])
    sig("aisx_freqsync_estimate_ahead", i32, [vp, vp, lng, i32, vp, vp])
    sig("aisx_agc_create", i32, [pvp, i32, f32, i32, i32])
    sig("aisx_agc_destroy", i32, [vp])
    sig("aisx_agc_geometry", i32, [vp, pi32, pi32, pi32, pi32])
    sig("aisx_agc_reset", i32, [vp])
    sig("aisx_agc_set_floor", i32, [vp, f32])
    sig("aisx_agc_set_streaming", i32, [vp, i32])
    sig("aisx_agc_set_lds_claim", i32, [vp, i32])
    sig("aisx_agc_get_lds_claim", i32, [vp, pi32, pi32])
    sig("aisx_freqsync_set_walk_lds_claim", i32, [vp, i32])
    sig("aisx_freqsync_get_walk_lds_claim", i32, [vp, pi32, pi32])
    sig("aisx_msk_placement", i32, [vp, pi32, pi32])
    sig("aisx_agc_process", i32, [vp, vp, lng, vp, lng, i32, vp])
    sig("aisx_chain_create", i32, [pvp, vp, vp, vp, vp, i32, i32, i32])
    sig("aisx_chain_destroy", i32, [vp])
    sig("aisx_chain_depth", i32, [])
    sig("aisx_chain_step", i32, [vp, vp, lng, i32, vp, lng, i32, vp, vp, lng, vp, vp, C.POINTER(C.c_longlong)])
    sig("aisx_chain_wait", i32, [vp, C.c_longlong, vp, i32])
    sig("aisx_chain_wait_input", i32, [vp, C.c_longlong, vp, i32])
    sig("aisx_chain_synchronize", i32, [vp])
    sig("aisx_chain_read_corr_output", i32, [vp, C.c_longlong, i32, i32, vp, lng, pi32, vp])
    sig("aisx_chain_read_tags", i32, [vp, C.c_longlong, vp, i32, pi32, vp])
    sig("aisx_chain_stream", vp, [vp, i32])
    sig("aisx_pfb_create", i32, [pvp, i32, i32, vp, i32, i32, i32])
    sig("aisx_pfb_destroy", i32, [vp])
    sig("aisx_pfb_process", i32, [vp, vp, lng, i32, vp, lng, pi32, vp])
    sig("aisx_hdlc_create", i32, [pvp, i32, i32])
    sig("aisx_hdlc_destroy", i32, [vp])
    sig("aisx_hdlc_work", i32, [vp, vp, i32, vp, i32, vp, i32, pi32])
    sig("aisx_pdu_to_nmea", i32, [C.c_char_p, vp, i32, C.c_char_p, i32])
    _lib = L
    return L


def check(rc, what=""):
    """Map a C-ABI status to the exception the reference's Python binding raises."""
    if rc >= 0:
        return rc
    msg = lib().aisx_last_error().decode("utf-8", "replace")
    if rc == AISX_ERR_OUT_OF_RANGE:
        raise IndexError(msg or what)  # SWIG maps std::out_of_range to IndexError
    if rc == AISX_ERR_NO_DEVICE:
        raise NoDeviceError(msg or "no HIP device")
    if rc == AISX_ERR_RUNTIME:
        raise RuntimeError(msg or what)  # std::runtime_error
    if rc == AISX_ERR_INVALID:
        raise ValueError("%s: %s" % (what, msg))
    if rc == AISX_ERR_OVERFLOW:
        raise OverflowError("%s: %s" % (what, msg))
    raise AisxError("%s failed (%d): %s" % (what, rc, msg))
