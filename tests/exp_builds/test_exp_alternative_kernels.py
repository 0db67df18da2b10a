"""-m gpu twin tests of the ALTERNATIVE kernels, which only lib/libaisx_exp.so (the -DAISX_EXPERIMENTS build) contains and
only it selects through AISX_* environment variables: the correlator's predecessors (round 1's plain-load kernels, the
256-thread and the 512-thread LDS-DMA builds) against the product's kernel and the oracle, and the timing recovery at 4 .. 64
channels per wave.  They run in a process of their own (tests/test_gpu_exp_builds.py starts it with AISX_LIB_VARIANT=exp);
collected in a process that loaded the product library they skip.
"""
import os

import numpy as np
import pytest

import oracle_py as orc
from parity import assert_tags_match, planted, unit_template

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("AISX_LIB_VARIANT") != "exp", reason="needs lib/libaisx_exp.so: run through tests/test_gpu_exp_builds.py")]


@pytest.fixture(scope="module")
def ais():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a visible MI355X"
    import ais_amd
    from ais_amd import _lib

    assert _lib.LIB_PATH.endswith("libaisx_exp.so")
    return ais_amd


def _dev(x):
    import torch

    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def _per_chan(tags, nchan):
    return [tags[tags["chan"] == c] for c in range(nchan)]


@pytest.mark.parametrize("N", [20, 112, 512, 896, 1500, 2048])
def test_corr_builds_agree(ais, N, monkeypatch):
    # AISX_CORR_DMA=0 selects round 1's correlators (k_corr_main, k_corr4_main: plain window loads, H and twiddles from L2);
    # AISX_CORR_WIDE=0 / 1 the F = 4096 builds k_corr4d_main (256 threads x 16 points, LDS-DMA) and k_corr4e_main (512 threads
    # x 8 points, LDS-DMA) -- the A/B partners of what the product runs (k_corr2d_main, k_corr4f_main).  Same contract: the
    # same input through every build gives the same pass-through bits, the same tags to the tolerance of two FFT orderings,
    # and all match the oracle.
    rng = np.random.default_rng(300 + N)
    tmpl = unit_template(rng, N)
    lens = [9000, N // 2 + 1, 7000]
    pos = [[700, 2900, 9000 - N // 2], [5, 12000], []]
    x = planted(rng, 3, sum(lens), tmpl, pos)
    blks = {}
    for name, env in (("round1", {"AISX_CORR_DMA": "0"}), ("4d", {"AISX_CORR_WIDE": "0"}), ("4e", {"AISX_CORR_WIDE": "1"}), ("product", {})):
        if N <= 512 and name in ("4d", "4e"):
            continue  # (F = 2048: one LDS-DMA build, the product's)
        monkeypatch.delenv("AISX_CORR_DMA", raising=False)
        monkeypatch.delenv("AISX_CORR_WIDE", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        blks[name] = ais.corr_est_cc(tmpl, 4.0, 1, 0.9, nchan=3, max_items=max(lens), max_tags_per_chan=512)
    ora = [orc.CorrEst(tmpl, 4.0, 1, 0.9) for _ in range(3)]
    k = ndet = 0
    for L in lens:
        xd = _dev(x[:, k:k + L])
        outs = {name: b.work(xd)[0].cpu().numpy() for name, b in blks.items()}
        tags = {name: _per_chan(b.tags(), 3) for name, b in blks.items()}
        for c in range(3):
            want_out, _, want_tags = ora[c].work(x[c, k:k + L])
            for name in blks:
                assert np.array_equal(outs[name][c].view(np.uint32), want_out.view(np.uint32)), (name, c)
                nd = assert_tags_match(tags[name][c], want_tags)
            ndet += nd
        k += L
    assert ndet >= 4


@pytest.mark.parametrize("lpw", [4, 8, 16, 32, 64])
def test_msk_channels_per_wave_builds(ais, lpw, monkeypatch):
    # the three builds of the timing-recovery kernel (16 / 32 / 64 channels per wave) give the
    # same bits and symbols as the oracle; the library picks 16, AISX_MSK_LPW overrides it
    import synth

    monkeypatch.setenv("AISX_MSK_LPW", str(lpw))
    rng = np.random.default_rng(40 + lpw)
    nchan, lens = 70, [5000, 3000]
    total = sum(lens)
    xs = np.stack([synth.make_channel(800 + c, total, "P", 4, amp=1.0, cfo_max=50.0)[0] for c in range(nchan)])
    blk = ais.msk_timing_recovery_cc(4.0, 0.04, 0.01, 1, nchan=nchan, max_items=max(lens))
    cap = 32
    tg_all = []
    for c in range(nchan):
        t = np.zeros(12, dtype=ais.TAG_DTYPE)
        t["offset"] = np.sort(rng.choice(np.arange(10, total - 10), size=12, replace=False))
        t["value"] = rng.uniform(-0.9, 0.9, 12)
        t["key"] = 2
        t["chan"] = c
        tg_all.append(t)
    o = [orc.MskStream(4.0, 0.04, 0.01, 1) for _ in range(nchan)]
    import torch
    k = 0
    for L in lens:
        tg = np.zeros((nchan, cap), dtype=ais.TAG_DTYPE)
        cnt = np.zeros(nchan, np.int32)
        sels = []
        for c in range(nchan):
            sel = tg_all[c][(tg_all[c]["offset"] >= k) & (tg_all[c]["offset"] < k + L)]
            tg[c, : len(sel)] = sel
            cnt[c] = len(sel)
            sels.append(sel)
        d_tags = torch.as_tensor(tg.view(np.uint8).reshape(nchan, -1).copy()).cuda()
        d_cnt = torch.as_tensor(cnt).cuda()
        r = blk.work(_dev(xs[:, k:k + L]), tags_ptrs=(d_tags.data_ptr(), d_cnt.data_ptr(), cap))
        assert blk.last_status() == 0
        prod = r["produced"].cpu().numpy()
        syms = r["syms"].cpu().numpy()
        for c in range(0, nchan, 3):
            ot = np.zeros(len(sels[c]), dtype=orc.TAG_DTYPE)
            ot["offset"], ot["value"], ot["key"] = sels[c]["offset"], sels[c]["value"], sels[c]["key"]
            out, _, _, _ = o[c].step(xs[c, k:k + L], ot, want_aux=True)
            assert prod[c] == len(out)
            assert np.array_equal(syms[c, :prod[c]].view(np.uint32), out.view(np.uint32))
        for c in range(nchan):  # keep the oracles of the channels not compared in step
            if c % 3:
                ot = np.zeros(len(sels[c]), dtype=orc.TAG_DTYPE)
                ot["offset"], ot["value"], ot["key"] = sels[c]["offset"], sels[c]["value"], sels[c]["key"]
                o[c].step(xs[c, k:k + L], ot, want_aux=True)
        k += L
