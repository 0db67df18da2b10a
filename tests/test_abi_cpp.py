"""A compiled caller of the C ABI and BASELINE config 1.

tests/abi_cpp/sched_harness.cpp drives the four blocks of python/ais_demod.py:56 for ONE channel
through libaisx.so's *_work_host entry points only, the way the GNU Radio scheduler would (history,
output multiple, <= 24576 items per call, forecast back-off, tag store) under the deterministic
policy of tests/sched_policy.py, and checks bits and tags against tests/golden/config1_sched.bin
(the CPU oracle under the same policy).  -m "not gpu": the oracle still reproduces the fixture and
the harness compiles and links against the library; -m gpu: it runs."""
import os
import subprocess

import numpy as np
import pytest

import sched_policy as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "config1_sched.bin")
EXE = os.path.join(ROOT, "tests", "abi_cpp", "sched_harness")
CHAIN_EXE = os.path.join(ROOT, "tests", "abi_cpp", "chain_harness")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "gr-ais_amd"), "-s"])
    lib = os.path.join(ROOT, "gr-ais_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "abi_cpp", "sched_harness.cpp"), "-L", lib, "-laisx",
                           "-Wl,-rpath," + lib, "-o", EXE])
    return EXE


def test_config1_fixture_is_what_the_oracle_produces():
    fx = sp.read_fixture(FIXTURE)
    r = sp.run(sp.OracleBlocks(fx["tmpl"], float(fx["sps"])), fx["x"])
    assert np.array_equal(r["bits"], fx["bits"])
    assert len(r["tags"]) == len(fx["tags"]) > 100
    got = np.array(r["tags"], dtype=[("offset", "<u8"), ("value", "<f8"), ("key", "<i4")])
    for k in ("offset", "value", "key"):
        assert np.array_equal(got[k], fx["tags"][k])
    # the scheduler model really chunks: several corr_est and general_work calls of different sizes
    calls = np.array(r["calls"])
    assert len(calls) > 50 and len(set(calls[:, 0])) >= 4 and calls[:, 2].sum() > 40000
    assert len(fx["bursts"]) >= 5


def test_harness_compiles_against_the_header_and_library():
    exe = _build()
    assert os.path.exists(exe)
    syms = subprocess.check_output(["nm", "-D", "--undefined-only", exe], text=True)
    used = sorted({ln.split()[-1] for ln in syms.splitlines() if " aisx_" in ln})
    # the GNU Radio path only: no batched (device pointer) entry point is referenced
    assert {"aisx_corr_work_host", "aisx_msk_general_work_host", "aisx_agc_work_host", "aisx_freqsync_work_host",
            "aisx_msk_forecast", "aisx_corr_output_multiple", "aisx_corr_history"} <= set(used)
    assert not [u for u in used if u.endswith("_process") or u.endswith("_process_stream")], used


@pytest.mark.gpu
def test_config1_through_the_compiled_caller():
    exe = _build()
    out = subprocess.run([exe, FIXTURE], capture_output=True, text=True, timeout=600)
    print(out.stdout[-1500:], out.stderr[-1500:])
    assert out.returncode == 0 and "PASS" in out.stdout


def _build_chain():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "gr-ais_amd"), "-s"])
    lib = os.path.join(ROOT, "gr-ais_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I",
                           os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_cpp", "chain_harness.cpp"), "-L", lib,
                           "-laisx", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", CHAIN_EXE])
    return CHAIN_EXE


def test_chain_harness_compiles_and_uses_the_chain_entry_points():
    """tests/abi_cpp/chain_harness.cpp: the pipelined chain (aisx_chain_*) from a compiled C++ caller, device
    buffers of its own (plain HIP runtime), against the four stage calls made one after the other."""
    exe = _build_chain()
    syms = subprocess.check_output(["nm", "-D", "--undefined-only", exe], text=True)
    used = {ln.split()[-1] for ln in syms.splitlines() if " aisx_" in ln}
    assert {"aisx_chain_create", "aisx_chain_step", "aisx_chain_wait", "aisx_chain_wait_input", "aisx_chain_synchronize",
            "aisx_chain_destroy", "aisx_corr_read_tags_back", "aisx_freqsync_process", "aisx_agc_process", "aisx_corr_process",
            "aisx_msk_process_stream"} <= used
    # without a device it says so and fails: the library has no CPU path
    import ctypes as C

    n = C.c_int(0)
    C.CDLL(os.path.join(ROOT, "gr-ais_amd", "lib", "libaisx.so")).aisx_device_count(C.byref(n))
    if n.value <= 0:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert out.returncode == 3 and "no device" in out.stderr


@pytest.mark.gpu
def test_pipelined_chain_through_the_compiled_caller():
    exe = _build_chain()
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    print(out.stdout[-1500:], out.stderr[-1500:])
    assert out.returncode == 0 and "PASS" in out.stdout
