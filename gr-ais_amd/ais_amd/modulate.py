"""Host-side template generation: the MI355X build's counterpart of
`digital.gmsk_mod(sps, bt)` driven by `digital.modulate_vector_bc(mod, data,
taps)` (python/ais_demod.py:36-38; contract documented in the reference at
include/ais/modulate_vector.h:48-65).  Runs once at construction; the result is
an *input* of corr_est_cc, so it only has to be a correct GMSK modulator.
"""
import numpy as np


class gmsk_mod:
    """digital.gmsk_mod(samples_per_symbol, bt): packed bytes in, GMSK out.
    packed_to_unpacked(MSB first) -> {0,1}->{-1,+1} -> interpolating FIR with
    convolve(gaussian(1, sps, bt, 4*sps), ones(sps)) -> FM with pi/2/sps."""

    def __init__(self, samples_per_symbol=2, bt=0.35):
        sps = int(samples_per_symbol)
        if sps < 2 or sps != samples_per_symbol:
            raise TypeError("samples_per_symbol must be an integer >= 2, is %r" % (samples_per_symbol,))
        self.sps, self.bt = sps, float(bt)
        ntaps = 4 * sps
        t = np.arange(ntaps, dtype=np.float64) + 1 - 0.5 * ntaps
        s = 1.0 / (np.sqrt(np.log(2.0)) / (2 * np.pi * self.bt))
        g = np.exp(-0.5 * (s * t / sps) ** 2).astype(np.float32).astype(np.float64)
        g = (g / g.sum()).astype(np.float32).astype(np.float64)
        self.taps = np.convolve(g, np.ones(sps)).astype(np.float32)
        self.sensitivity = np.float32((np.pi / 2) / sps)

    def to_basic_block(self):
        return self

    def modulate(self, data):
        data = np.asarray(data, dtype=np.uint8)
        bits = np.unpackbits(data)  # MSB first
        nrz = bits.astype(np.float32) * 2 - 1
        up = np.zeros(nrz.size * self.sps, dtype=np.float32)
        up[:: self.sps] = nrz
        f = np.convolve(up.astype(np.float64), self.taps.astype(np.float64))[: up.size]
        ph = np.cumsum(f * np.float64(self.sensitivity))
        return np.exp(1j * ph).astype(np.complex64)


def modulate_vector_bc(modulator, data, taps):
    """digital.modulate_vector_bc(modulator, data, taps): modulate `data` and
    apply the post-modulation shaping filter `taps` (fir_filter_ccf(1, taps))."""
    y = modulator.to_basic_block().modulate(data)
    taps = np.asarray(taps, dtype=np.float32)
    if taps.size == 1 and taps[0] == 1.0:
        return y
    return np.convolve(y, taps)[: y.size].astype(np.complex64)
