"""N3 (BASELINE config 5): the polyphase channelizer against the reference's
per-channel freq_xlating_fir_filter_ccf (oracle restatement).  CPU lane model
here; the -m gpu test at the bottom runs the same comparison on the device."""
import numpy as np
import pytest

import oracle_py as orc

FS = 25e6


def _wideband(rng, n):
    # a few narrow-band tones/carriers on lane centres and between them, plus noise
    t = np.arange(n)
    x = 0.05 * (rng.normal(size=n) + 1j * rng.normal(size=n))
    for lane, amp, off in [(3, 1.0, 0.0), (200, 0.7, 3000.0), (777, 0.5, -5000.0), (1023, 0.9, 1000.0)]:
        f = lane * FS / 1024 + off
        x = x + amp * np.exp(2j * np.pi * f / FS * t + 1j * rng.uniform(-3, 3))
    return x.astype(np.complex64)


def _check(out, x, taps, decim, lanes, k_list):
    for m in lanes:
        for k in k_list:
            want = orc.freq_xlating_fir(taps, decim, m * FS / 1024, FS, x, k, 1)[0]
            got = out[m, k]
            assert abs(got - want) < 2e-4 * max(1.0, abs(want)), (m, k, got, want)


@pytest.mark.parametrize("decim", [1024, 512])
def test_emul_pfb_matches_freq_xlating_filter(decim):
    import emul_py as emu

    rng = np.random.default_rng(decim)
    taps = orc.firdes_low_pass(1.0, FS, 11e3, 1e3)
    assert taps.size == 60227 and abs(taps.sum() - 1.0) < 1e-3
    nframes = [6, 2, 5]
    x = _wideband(rng, sum(nframes) * decim)
    p = emu.Pfb(taps, decim)
    outs, k = [], 0
    for nf in nframes:  # state (history, frame counter) carries across calls
        outs.append(p.work(x[k:k + nf * decim]))
        k += nf * decim
    out = np.concatenate(outs, axis=1)
    _check(out, x, taps, decim, [0, 3, 200, 511, 777, 1023], [0, 1, 5, 7, 12])
    # the tone sitting on lane 3 shows up on that lane and not on a far one (the 60227-tap
    # prototype is 59 frames long, so by frame 12 it is still filling)
    assert abs(out[3, 12]) > 20 * abs(out[100, 12])


def test_host_firdes_matches_oracle():
    import ais_amd

    a = ais_amd.firdes_low_pass(1.0, 250e3, 11e3, 1e3)
    b = orc.firdes_low_pass(1.0, 250e3, 11e3, 1e3)
    assert a.size == b.size == 603
    assert np.max(np.abs(a - b)) < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("decim", [1024, 512])
def test_gpu_pfb_matches_freq_xlating_filter(decim):
    import torch

    import ais_amd

    rng = np.random.default_rng(decim + 1)
    taps = ais_amd.firdes_low_pass(1.0, FS, 11e3, 1e3)
    nframes = [64, 3, 40]
    x = _wideband(rng, sum(nframes) * decim)
    p = ais_amd.pfb_channelizer_ccf(1024, taps, decim=decim, max_frames=128)
    outs, k = [], 0
    for nf in nframes:
        outs.append(p.work(torch.as_tensor(x[k:k + nf * decim]).cuda()).cpu().numpy())
        k += nf * decim
    out = np.concatenate(outs, axis=1)
    _check(out, x, taps, decim, [0, 3, 200, 511, 777, 1023], [0, 1, 60, 65, 70, 106])
    with pytest.raises(ValueError):
        ais_amd.pfb_channelizer_ccf(512, taps)
