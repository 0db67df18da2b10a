import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'gr-ais_amd'))
import numpy as np, torch
import ais_amd as ais
import oracle_py as orc
from ais_amd import synth
sps=4
tmpl = ais.modulate_vector_bc(ais.gmsk_mod(sps, 0.4), [1, 1, 0, 0] * 7, [1])
nchan, T, steps = 24, 16384, 3
xs = np.stack([synth.make_channel(900 + c, T * steps, "S", sps, amp=1.0, cfo_max=3.0)[0] for c in range(nchan)])
opts = dict(samples_per_symbol=sps, bits_per_sec=9600.0, clockrec_gain=0.04, omega_relative_limit=0.01, fftlen=1024)
dem = ais.ais_demod(opts, nchan=nchan, max_items=T, stages="core", preamble_symbols=tmpl)
ora = [orc.Demod(sps, tmpl, stages=0) for _ in range(nchan)]
for s in range(steps):
    chunk = xs[:, s*T:(s+1)*T]
    r = dem.work(torch.as_tensor(chunk).cuda())
    prod = r["produced"].cpu().numpy()
    exp = np.array([len(ora[c].step(chunk[c])[0]) for c in range(nchan)])
    print(s, "status", dem.clockrec.last_status(), "mismatch at", np.nonzero(prod != exp)[0], prod[:6], exp[:6])
