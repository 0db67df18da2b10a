#!/usr/bin/env python3
"""Regenerates config1_sched.bin:  python tests/golden/make_config1.py

BASELINE config 1: ONE 4-sps channel through the stock python/ais_demod.py:56 flowgraph, the blocks
driven as the GNU Radio scheduler drives them (history, output multiple, <= 24576 items per call,
forecast back-off, tag store) by the deterministic scheduler model of tests/sched_policy.py, with
this repository's CPU oracle as the blocks.  Like every fixture here it is a REGRESSION vector
(inputs + the oracle's outputs), not a reference vector: see make_golden.py.  Flat little-endian
layout (read by tests/abi_cpp/sched_harness.cpp and by sched_policy.read_fixture)."""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "gr-ais_amd"))

import sched_policy as sp  # noqa: E402
from ais_amd import gmsk_mod, modulate_vector_bc  # noqa: E402
import synth  # noqa: E402  (tests/synth.py)


def main():
    sps, T = 4, 3 * 16384 + 700
    tmpl = modulate_vector_bc(gmsk_mod(sps, 0.4), [1, 1, 0, 0] * 7, [1])
    x, infos = synth.make_channel(9100, T, "S", sps, amp=0.3, cfo_max=500.0)
    r = sp.run(sp.OracleBlocks(tmpl, float(sps)), x)
    bits = r["bits"]
    tags = np.array(r["tags"], dtype=[("offset", "<u8"), ("value", "<f8"), ("key", "<i4")])
    bursts = []
    for inf in infos:
        pat = np.asarray(inf["data_bits"], np.uint8)
        for pos in synth.find_bits(bits, pat):
            bursts.append((pos, pat.size))
    bursts = np.array(bursts, np.int32).reshape(-1, 2)
    calls = np.array(r["calls"], np.int32)
    print("config 1: %d samples -> %d bits, %d tags (%d detections), %d msk calls, %d decoded bursts of %d sent"
          % (T, bits.size, len(tags), int((tags["key"] == 0).sum()), len(calls), len(bursts), len(infos)))
    with open(os.path.join(HERE, "config1_sched.bin"), "wb") as f:
        f.write(b"AISXC1\0\0")
        f.write(struct.pack("<8i", tmpl.size, T, len(tags), bits.size, len(bursts), len(sp.SRC_PIECES), len(sp.CORR_K),
                            len(sp.MSK_CAPS)))
        f.write(struct.pack("<f", float(sps)))
        f.write(tmpl.astype("<c8").tobytes())
        f.write(x.astype("<c8").tobytes())
        f.write(np.array(sp.SRC_PIECES, "<i4").tobytes())
        f.write(np.array(sp.CORR_K, "<i4").tobytes())
        f.write(np.array(sp.MSK_CAPS, "<i4").tobytes())
        rec = np.zeros(len(tags), dtype=[("offset", "<u8"), ("value", "<f8"), ("key", "<i4"), ("chan", "<i4")])
        rec["offset"], rec["value"], rec["key"] = tags["offset"], tags["value"], tags["key"]
        f.write(rec.tobytes())
        f.write(bits.tobytes())
        f.write(bursts.astype("<i4").tobytes())


if __name__ == "__main__":
    main()
