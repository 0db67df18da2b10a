"""The twin tests of the alternative kernels (tests/exp_builds/) need lib/libaisx_exp.so, which a process loads instead of the
product library when AISX_LIB_VARIANT=exp is set before the first call: this test starts that process."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_alternative_kernels_in_the_experiments_build():
    env = dict(os.environ, AISX_LIB_VARIANT="exp")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "exp_builds"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "skipped" not in tail.splitlines()[-1], tail


def test_product_library_reads_no_environment():
    """`strings libaisx.so | grep AISX_` is empty for the product build; the experiments build has the knobs."""
    lib = os.path.join(ROOT, "gr-ais_amd", "lib")
    for name, want in (("libaisx.so", False), ("libaisx_exp.so", True)):
        path = os.path.join(lib, name)
        if not os.path.exists(path):
            pytest.skip("library not built")
        out = subprocess.run(["strings", path], capture_output=True, text=True).stdout
        knobs = sorted({w for w in out.split() if re.fullmatch(r"AISX_[A-Z0-9_]+", w)})
        assert bool(knobs) == want, (name, knobs[:10])
