"""How often could msk_timing_recovery be restarted at a pair of time_est tags?

Runs the oracle's stock chain on the benchmark's kind of input (CPU copy of bench.make_input) with the
oracle's reset trace switched on and prints, per channel and step: tags in range, tags that reset the
loop, tags left unused when the call ended (the first of them blocks the rest, lib/
msk_timing_recovery_cc_impl.cc:140-142), and how many resets came exactly two iterations after the
previous one (the history-free restart points).  Diagnostic only.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "gr-ais_amd"))
sys.path.insert(0, ROOT)


def main():
    import oracle_py as orc
    import bench
    import synth

    nchan = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    T, sps, family = 65536, 4, "S"
    tmpl = bench.make_template(family, sps)
    ip = bench.input_params(family, True)
    L = orc.lib()
    L.orc_msk_set_trace.argtypes = [C.c_void_p, C.c_int]
    rng = np.random.default_rng(1)
    tot = np.zeros(6, dtype=np.int64)
    for c in range(nchan):
        x = synth.make_channel(synth.SEED0 + c, T, family, sps, amp=ip["amp"], cfo_max=ip["cfo_max"], noise=False)[0]
        sigma = ip["amp"] * np.sqrt(sps / (10 ** (20.0 / 10.0)) / 2.0)
        dem = orc.Demod(sps, tmpl, stages=3)
        for s in range(nsteps):
            xs = (x + (rng.standard_normal(T) + 1j * rng.standard_normal(T)) * sigma).astype(np.complex64)
            buf = np.zeros(4 * 100000, dtype=np.int32)
            L.orc_msk_set_trace(buf.ctypes.data, 100000)
            dem.step(xs)
            n = L.orc_msk_trace_count()
            L.orc_msk_set_trace(None, 0)
            r = buf[: 4 * n].reshape(n, 4)
            fired = r[r[:, 0] == 1]
            ends = r[r[:, 0] == 2]
            nin, nused = int(ends[:, 1].sum()), int(ends[:, 2].sum())
            two = int((fired[:, 2] == 2).sum())
            # restart points at least 1024 samples apart
            pts, last = 0, -10 ** 9
            for k in range(len(fired)):
                if fired[k, 2] == 2 and fired[k, 1] - last >= 2048:
                    pts += 1
                    last = fired[k, 1]
            print("ch %2d step %d: calls %d  tags in range (first call) %d  resets %d  unused at end of first call %d  "
                  "resets 2 iterations after a reset %d  restart points >= 2048 apart %d  gaps %s" %
                  (c, s, len(ends), ends[0, 1], len(fired), ends[0, 1] - ends[0, 2], two, pts,
                   np.bincount(np.minimum(fired[:, 2], 9))[:10].tolist()))
            tot += [1, ends[0, 1], len(fired), ends[0, 1] - ends[0, 2], two, pts]
    print("total: channel-steps %d, tags %d, resets %d, unused %d, two-apart %d, restart points %d" % tuple(tot))


if __name__ == "__main__":
    main()
