"""Diagnostic: the stock chain for a few pipelined steps on bench.py's input; prints ms per step and what the
time-parallel timing recovery (k_mskp.h) made of the last step.  usage: msk_tp_probe.py [nchan] [steps] [chain]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gr-ais_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch

    import ais_amd
    import bench

    nchan = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    chain = sys.argv[3] if len(sys.argv) > 3 else "stock"
    T, sps = 65536, 4
    dev = torch.device("cuda", 0)
    tmpl = bench.make_template("S", sps)
    opts = dict(samples_per_symbol=sps, bits_per_sec=9600.0, clockrec_gain=0.04, omega_relative_limit=0.01, fftlen=1024)
    x = bench.make_input(nchan, T, "S", sps, dev, 0, chain == "stock")
    dem = ais_amd.ais_demod(opts, nchan=nchan, max_items=T, stages=chain, preamble_symbols=tmpl, fused_front_end=True)
    for _ in range(3):
        dem.work_pipelined(x, x_next=x if chain == "stock" else None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        dem.work_pipelined(x, x_next=x if chain == "stock" else None)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    st = dem.clockrec.restart_stats()
    print("nchan %d chain %s: %.3f ms/step  status %d  %s" % (nchan, chain, ms, dem.clockrec.last_status(), st), flush=True)
    if st["calls"]:
        print("  per channel: %.1f restart points, %.1f units taken, %.1f %% of the symbols from units" %
              (st["restart_points"] / nchan, st["units_taken"] / nchan,
               100.0 * st["symbols_from_units"] / max(1, int(dem._chain_outs[0]["produced"].sum().item()))))


if __name__ == "__main__":
    main()
