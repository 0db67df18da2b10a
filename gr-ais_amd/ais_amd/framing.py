"""Host-side tail of the receive chain (python/radio.py:64-73): the HDLC deframer
`digital.hdlc_deframer_bp(11, 64)` and `ais.pdu_to_nmea(designator)`
(lib/pdu_to_nmea_impl.cc).  Per-packet work on the CPU, in libaisx.so's host
code; no GPU needed."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


class hdlc_deframer_bp:
    def __init__(self, length_min, length_max):
        h = C.c_void_p()
        check(_lib.lib(device=False).aisx_hdlc_create(C.byref(h), int(length_min), int(length_max)), "hdlc_deframer_bp")
        self._h = h
        self._max = int(length_max)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _lib.lib(device=False).aisx_hdlc_destroy(h)
            self._h = None

    def work(self, bits):
        """bits: unpacked bits (one per item).  Returns the list of PDUs (bytes) whose CRC checked."""
        b = np.ascontiguousarray(bits, dtype=np.uint8)
        maxp = b.size // 16 + 2
        buf = np.zeros(maxp * (self._max + 2), dtype=np.uint8)
        offs = np.zeros(maxp + 1, dtype=np.int32)
        n = C.c_int(0)
        check(_lib.lib(device=False).aisx_hdlc_work(self._h, b.ctypes.data_as(C.c_void_p), b.size, buf.ctypes.data_as(C.c_void_p),
                                        buf.size, offs.ctypes.data_as(C.c_void_p), maxp, C.byref(n)), "hdlc work")
        return [bytes(buf[offs[k]:offs[k + 1]]) for k in range(n.value)]


class pdu_to_nmea:
    def __init__(self, designator):
        self.designator = str(designator)

    def msg_to_sentence(self, pdu):
        p = np.frombuffer(bytes(pdu), dtype=np.uint8)
        out = C.create_string_buffer(4096)
        n = check(_lib.lib(device=False).aisx_pdu_to_nmea(self.designator.encode(), p.ctypes.data_as(C.c_void_p), p.size, out, 4096),
                  "pdu_to_nmea")
        return out.raw[:n].decode("latin-1")
