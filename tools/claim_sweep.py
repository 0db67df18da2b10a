#!/usr/bin/env python3
"""tools/claim_sweep.py [--nchan 1024,2048,...] [--claims 0,24,48,72] [--steps 20] [--reps 2]

The front-end kernel's LDS claim (aisx_agc_set_lds_claim: how many k_agcw workgroups fit on a CU beside a timing
recovery workgroup) against the channel count: ms per pipelined step of the stock chain, the chain's own choice
(aisx_chain_create) first, then every claim of the sweep set on the same chain object, interleaved `reps` times.
One JSON line per channel count; DESIGN_APPENDIX.md A.6 holds the table, tests/test_gpu_configs.py the gate
(the chain's choice within 2 % of the best of the sweep)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-ais_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def sweep(nchan, claims_kb, steps, reps, T=65536, sps=4):
    import numpy as np
    import torch

    import ais_amd
    import bench
    from ais_amd import _lib

    device = torch.device("cuda:0")
    tmpl = bench.make_template("S", sps)
    x = bench.make_input(nchan, T, "S", sps, device, 0, True)
    opts = dict(samples_per_symbol=sps, bits_per_sec=9600.0, clockrec_gain=0.04, omega_relative_limit=0.01, fftlen=1024)
    dem = ais_amd.ais_demod(opts, nchan=nchan, max_items=T, stages="stock", preamble_symbols=tmpl, fused_front_end=True)
    L = _lib.lib()
    dem.work_pipelined(x, x_next=x)  # (the chain is made by the first step)
    torch.cuda.synchronize()
    chosen = C.c_int(-1)
    L.aisx_agc_get_lds_claim(dem.agc._h, C.byref(chosen), None)

    def run():
        for _ in range(4):
            dem.work_pipelined(x, x_next=x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            dem.work_pipelined(x, x_next=x)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    bench.spin_up(torch, device)
    res = {}
    order = [("chosen", chosen.value)] + [("%d" % k, k * 1024) for k in claims_kb]
    _lib.check(L.aisx_agc_set_lds_claim(dem.agc._h, chosen.value), "set_lds_claim")
    run()  # (untimed: clocks, caches)
    for r in range(reps):
        for name, claim in order[r % len(order):] + order[:r % len(order)]:  # (a different first one every time round)
            _lib.check(L.aisx_agc_set_lds_claim(dem.agc._h, claim), "set_lds_claim")
            res.setdefault(name, []).append(run())
    _lib.check(L.aisx_agc_set_lds_claim(dem.agc._h, chosen.value), "set_lds_claim")
    out = dict(nchan=nchan, chosen_claim_bytes=chosen.value, steps=steps,
               ms_per_step={k: round(float(np.median(v)), 4) for k, v in res.items()},
               ms_per_step_all={k: [round(t, 4) for t in v] for k, v in res.items()})
    best = min(v for k, v in out["ms_per_step"].items() if k != "chosen")
    out["best_of_sweep_ms"] = best
    out["chosen_over_best"] = round(out["ms_per_step"]["chosen"] / best, 4)
    del dem, x
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--nchan", default="1024,2048,3072,4096,6144,8192,12288,16384")
    ap.add_argument("--claims", default="0,24,48,72")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    for n in [int(v) for v in a.nchan.split(",")]:
        print(json.dumps(sweep(n, [int(v) for v in a.claims.split(",")], a.steps, a.reps)), flush=True)
