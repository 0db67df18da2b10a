"""The C-ABI shared library loads and exports every entry point that
include/aisx.h declares; argument validation and the reference's error
behaviour work without a GPU; the host-side helpers agree with the oracle.
(No compute calls here: -m "not gpu".)"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    so = os.path.join(ROOT, "gr-ais_amd", "lib", "libaisx.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "gr-ais_amd"), "-s"])
    assert os.path.exists(so)
    return so


def _declared():
    txt = open(os.path.join(ROOT, "include", "aisx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(aisx_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(libpath):
    L = C.CDLL(libpath)
    names = _declared()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    L.aisx_version.restype = C.c_int
    assert L.aisx_version() == 300


def test_binding_matches_header(libpath):
    from ais_amd import _lib

    L = _lib.lib()
    for n in _declared():
        assert getattr(L, n).argtypes is not None, n


def test_error_behaviour_without_device(libpath):
    import ais_amd
    from ais_amd import _lib

    n = C.c_int(-1)
    _lib.lib().aisx_device_count(C.byref(n))
    # std::out_of_range paths are checked before the device is touched (impl :61,:82)
    with pytest.raises(IndexError):
        ais_amd.msk_timing_recovery_cc(4.0, 0.0, 0.01, 1)
    with pytest.raises(IndexError):
        ais_amd.msk_timing_recovery_cc(4.0, 0.04, 0.01, 0)
    with pytest.raises(ValueError):
        ais_amd.corr_est_cc(np.ones(4000, np.complex64), 4.0, 1)
    with pytest.raises(ValueError):
        ais_amd.square_and_fft_sync_cc(38400.0, 9600.0, 512)
    with pytest.raises(ValueError):
        ais_amd.feedforward_agc_cc(0, 2.0)
    if n.value <= 0:
        # no GPU here: the product must fail loudly, there is no CPU path
        with pytest.raises(_lib.NoDeviceError):
            ais_amd.corr_est_cc(np.ones(16, np.complex64), 4.0, 1)
        with pytest.raises(_lib.NoDeviceError):
            ais_amd.msk_timing_recovery_cc(4.0, 0.04, 0.01, 1)


def test_product_never_touches_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "gr-ais_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"oracle_py|ais_oracle|libais_oracle|orc_", txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_template_generator_against_oracle():
    import oracle_py as orc
    from ais_amd import gmsk_mod, modulate_vector_bc

    for sps in (4, 5):
        a = modulate_vector_bc(gmsk_mod(sps, 0.4), [1, 1, 0, 0] * 7, [1])
        b = orc.gmsk_modulate_vector(sps, 0.4, [1, 1, 0, 0] * 7)
        assert a.size == b.size == 224 * sps
        assert np.max(np.abs(a - b)) < 5e-4
    with pytest.raises(TypeError):
        gmsk_mod(4.5, 0.4)
