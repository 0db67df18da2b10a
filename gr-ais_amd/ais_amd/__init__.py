"""ais_amd -- MI355X-native drop-in for the gr-ais per-sample demod hot path.

Mirrors the reference's Python-visible interface for this path
(python/__init__.py:29-32 + swig/ais_swig.i:16-23): `corr_est_cc`, `freqest`,
`msk_timing_recovery_cc`, the hier blocks `square_and_fft_sync_cc`
(python/gmsk_sync.py) and `ais_demod` (python/ais_demod.py), and the template
helper `modulate_vector_bc`.  All signal processing runs in libaisx.so (hand
written HIP for gfx950, C ABI in include/aisx.h).
"""
import os as _os

# the pipelined chain (ais_demod.work_pipelined, aisx_chain_*) keeps four streams busy; with the HIP
# runtime's default of four hardware queues two of them would share one and run in turn.  Read by
# the runtime at its first call, i.e. after this import.
import sys as _sys

_t = _sys.modules.get("torch")
# (a process that has already made HIP calls -- torch.cuda initialised -- keeps the queue count it started with:
# ais_demod.work_pipelined warns once when that is the case)
HW_QUEUES_SET_TOO_LATE = "GPU_MAX_HW_QUEUES" not in _os.environ and _t is not None and _t.cuda.is_initialized()
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
del _t

from .framing import hdlc_deframer_bp, pdu_to_nmea  # noqa: F401
from .modulate import gmsk_mod, modulate_vector_bc  # noqa: F401


def __getattr__(name):
    # blocks need torch + libaisx.so; load them on first use so that the host-only
    # helpers (modulate, framing) stay importable anywhere
    if name in ("corr_est_cc", "msk_timing_recovery_cc", "square_and_fft_sync_cc", "freqest", "feedforward_agc_cc",
                "ais_demod", "TAG_DTYPE", "pfb_channelizer_ccf", "firdes_low_pass", "freq_sync_agc"):
        from . import blocks

        return getattr(blocks, name)
    raise AttributeError(name)
