"""front-end kernels with the chip to themselves (run under rocprofv3 --kernel-trace --stats)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-ais_amd")); sys.path.insert(0, ROOT)
import torch
import ais_amd, bench
nchan, T = 4096, 65536
dev = torch.device("cuda", 0)
x = bench.make_input(nchan, T, "S", 4, dev, 0, True)
fs = ais_amd.square_and_fft_sync_cc(38400.0, 9600.0, 1024, nchan=nchan, max_items=T)
agc = ais_amd.feedforward_agc_cc(512, 2.0, nchan=nchan, max_items=T + 1024)
out = torch.empty((nchan, T + 1024), dtype=torch.complex64, device=dev)
for _ in range(8):
    y = ais_amd.freq_sync_agc(fs, agc, x, out=out)[0]
    torch.cuda.synchronize()
print("done", y.shape)
