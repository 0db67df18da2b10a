"""Random ragged streams through the time-parallel timing recovery on the device against the oracle (bit-exact or it
prints FAILED): call lengths, join kernel, max_noutput_items, share of failing junctions, tag density, a NaN tag, sps are
drawn per seed.  `gpurun -- python tools/fuzz_time_parallel.py`; 24 seeds take ~10 s.  Uses tests/test_gpu_mskp.py:_stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gr-ais_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import ais_amd
import test_gpu_mskp as T
bad = 0
for seed in range(100, 124):
    rng = np.random.default_rng(seed)
    lens = [int(x) for x in rng.integers(1, 30000, size=int(rng.integers(2, 6)))]
    join = int(seed % 2)
    Q = int(rng.choice([0, 64, 256, 1000]))
    neg = float(rng.choice([0.1, 0.5, 0.9]))
    every = int(rng.choice([300, 700, 1500]))
    sps = float(rng.choice([4.0, 4.0, 5.2083]))
    try:
        T._stream(ais_amd, 24, lens, seed=seed, join=join, Q=Q, neg_frac=neg, pair_every=every, nan_at=int(rng.integers(3, 40)), sps=sps)
        print("seed", seed, "ok", lens, join, Q, neg, every, sps, flush=True)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED", lens, join, Q, neg, every, sps, e, flush=True)
print("failures:", bad)
