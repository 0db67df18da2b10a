#!/usr/bin/env python3
"""A/B runs of bench.py with alternative builds of libaisx.so on ONE GPU box (box-to-box the
step time moves by about 1 %, more than most kernel changes are worth):

    hipcc ... -o tools/scratch/libaisx_a.so ...; hipcc ... -o tools/scratch/libaisx_b.so ...
    gpurun -- 'for v in a b a b; do python tools/ab_bench.py tools/scratch/libaisx_$v.so --no-cpu-baseline | tail -1; done'

Everything after the library path goes to bench.py unchanged."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-ais_amd"))
import ais_amd._lib as L  # noqa: E402

L.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
