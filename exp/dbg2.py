import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'gr-ais_amd'))
import numpy as np, torch
import ais_amd as ais
import oracle_py as orc
from ais_amd import synth
sps=4
lv = [1 if b else -1 for b in synth.sync_bits("P")]
tmpl = synth.gmsk_waveform(np.array(lv, float), sps)[: len(lv) * sps].astype(np.complex64)
nchan, T = 64, 65536
base = np.stack([synth.make_channel(4000 + c, T, "P", sps, amp=1.0, cfo_max=15.0)[0] for c in range(16)])
x = torch.as_tensor(base).cuda().repeat(nchan // 16, 1).contiguous()
for cap in (512, 2048):
    blk = ais.corr_est_cc(tmpl, 4.0, 1, 0.9, nchan=nchan, max_items=T, max_tags_per_chan=cap)
    msk = ais.msk_timing_recovery_cc(4.0, 0.04, 0.01, 1, nchan=nchan, max_items=T)
    out, _ = blk.work(x)
    r = msk.work(out, tags_from=blk, want_syms=False)
    prod = r["produced"].cpu().numpy()
    exp = np.array([len(orc.Demod(sps, tmpl, stages=0).step(base[c])[0]) for c in range(16)])
    print("cap", cap, "mismatch", np.nonzero(prod[:16] != exp)[0], prod[:4], exp[:4])
