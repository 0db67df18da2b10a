"""Pins for the CPU oracle.

The reference ships no tests (lib/qa_ais.cc:30-36 is empty), so the only known
answers are the control-flow observations the survey recorded from the
reference's own block sources.  Everything here runs on
the CPU (-m "not gpu").
"""
import os

import numpy as np
import pytest

import oracle_py as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tagdict(tags):
    return [(orc.KEY_NAMES[int(t["key"])], int(t["offset"]), float(t["value"])) for t in tags]


def test_freqest_kat_survey_4_1():
    # make(38400, 9600, 1024); X[394]=5+0j, X[650]=0+4j -> 187.5 Hz; the all-zero
    # second vector repeats it (stale maxpos, lib/freqest_impl.cc:68 vs :74)
    fe = orc.FreqEst.make(38400.0, 9600, 1024)
    assert fe.offset == 256 and fe.binsize == 37.5
    v = np.zeros((2, 1024), dtype=np.complex64)
    v[0, 394] = 5
    v[0, 650] = 4j
    out = fe.work(v)
    assert out.tolist() == [187.5, 187.5]
    # first vector of a work call all zero -> (0-512)*37.5/2 = -9600 Hz
    assert fe.work(np.zeros((1, 1024), np.complex64)).tolist() == [-9600.0]


def test_freqest_offsets():
    assert orc.FreqEst.make(48000.0, 9600, 1024).offset == 204
    assert orc.FreqEst.make(50000.0, 9600, 1024).offset == 196


def test_corr_est_kat_survey_4_1():
    rng = np.random.default_rng(7)
    N = 20
    tmpl = np.exp(1j * rng.uniform(-np.pi, np.pi, N)).astype(np.complex64)
    ce = orc.CorrEst(tmpl, 4.0, 1, 0.9)
    assert ce.history - 1 == 20
    assert ce.fftsize == 64
    assert ce.output_multiple == 45  # 64 - 20 + 1
    assert abs(ce.threshold - 0.9 * 400) < 1e-3
    x = np.zeros(360, dtype=np.complex64)
    x[100:120] = 1j * tmpl
    out, _, tags = ce.work(x)
    # output = input delayed by N
    assert np.array_equal(out[N:], x[:-N]) and np.all(out[:N] == 0)
    td = _tagdict(tags)
    assert [t[0] for t in td] == ["corr_start", "phase_est", "time_est", "corr_est"]
    assert [t[1] for t in td] == [119, 120, 120, 120]
    assert abs(td[0][2] - 400) < 1e-2 and abs(td[3][2] - 400) < 1e-2
    assert abs(td[1][2] - np.pi / 2) < 1e-3
    assert abs(td[2][2]) < 1e-5
    # sync word starts at output index k+1: out[P+N] == x[P]
    assert out[100 + N] == x[100]


def test_corr_est_chunk_boundary_dependence():
    # the centre of mass is 0 when the peak is the last
    # item of a work call, non-zero otherwise.
    rng = np.random.default_rng(11)
    N = 20
    tmpl = np.exp(1j * rng.uniform(-np.pi, np.pi, N)).astype(np.complex64)
    x = (rng.normal(0, 0.25 / np.sqrt(2), 360) + 1j * rng.normal(0, 0.25 / np.sqrt(2), 360)).astype(np.complex64)
    x[25:45] += tmpl
    a = orc.CorrEst(tmpl, 4.0, 1, 0.9)
    _, _, t1 = a.work(x)
    b = orc.CorrEst(tmpl, 4.0, 1, 0.9)
    t2 = []
    for k in range(0, 360, 45):
        t2.extend(b.work(x[k:k + 45])[2])
    d1 = {(t[0], t[1]): t[2] for t in _tagdict(t1)}
    d2 = {(t[0], t[1]): t[2] for t in _tagdict(t2)}
    assert ("time_est", 45) in d1 and ("time_est", 45) in d2
    assert d1[("time_est", 45)] != 0.0
    assert d2[("time_est", 45)] == 0.0
    assert abs(d1[("corr_start", 44)] - d2[("corr_start", 44)]) < 1e-3 * d1[("corr_start", 44)]


def test_corr_est_fft_matches_direct_form():
    rng = np.random.default_rng(3)
    for N in (20, 112, 896):
        tmpl = np.exp(1j * rng.uniform(-np.pi, np.pi, N)).astype(np.complex64)
        ce = orc.CorrEst(tmpl, 4.0, 1, 0.9)
        n = ce.output_multiple * 3
        x = (rng.normal(size=n) + 1j * rng.normal(size=n)).astype(np.complex64)
        _, corr, _ = ce.work(x, want_corr=True)
        taps = np.conj(tmpl[::-1]).astype(np.complex128)
        ref = np.convolve(x.astype(np.complex128), taps)[:n]
        assert np.max(np.abs(corr - ref)) < 2e-5 * np.sqrt(N) * np.max(np.abs(ref)) / np.sqrt(N)
        assert np.array_equal(ce.taps(), np.conj(tmpl[::-1]))


def test_msk_kat_survey_4_1():
    m = orc.Msk(4.0, 0.04, 0.01, 1)
    assert m.forecast(512) == 2062
    rng = np.random.default_rng(5)
    x = np.exp(1j * np.cumsum(rng.choice([-1.0, 1.0], 40000).repeat(1) * np.pi / 8)).astype(np.complex64)
    buf = np.concatenate([np.zeros(1, np.complex64), x])
    # first call / steady state input consumption for 512 outputs
    out, _, _, cons, st = m.general_work(512, 2062, buf, 1, np.zeros(0, orc.TAG_DTYPE), 0)
    assert len(out) == 512 and st == 0
    assert 2040 <= cons <= 2056
    # time_est tag at absolute offset 1000, value -0.25 -> mu = 0.75 at that symbol
    m2 = orc.Msk(4.0, 0.04, 0.01, 1)
    tags = np.zeros(1, orc.TAG_DTYPE)
    tags[0] = (1000, -0.25, orc.KEY_TIME_EST, 0)
    out, err, mu, cons, st = m2.general_work(400, 2000, buf, 1, tags, 0, want_aux=True)
    k = int(np.argmin(np.abs(mu - 0.75)))
    assert 248 <= k <= 252 and abs(mu[k] - 0.75) < 1e-6


def test_msk_ctor_errors():
    with pytest.raises(IndexError):
        orc.Msk(4.0, 0.0, 0.01, 1)
    with pytest.raises(IndexError):
        orc.Msk(4.0, 0.04, 0.01, 3)


def test_mmse_table_rows_recalled_from_upstream():
    # rows 1-3 and 64 of gr-filter interpolator_taps.h as recalled from upstream
    # (SURVEY 8c item 1); the generated table must reproduce them digit for digit
    import os
    import re
    txt = open(os.path.join(orc.ORACLE_DIR, "orc_tables.h")).read()
    rows = re.findall(r"\{ ([^}]*) \}, /\*\s*(\d+)/128", txt)
    tab = {int(r): [float(v.rstrip("f")) for v in vals.split(",")] for vals, r in rows}
    assert len(tab) == 129
    assert tab[0] == [0, 0, 0, 0, 1, 0, 0, 0] and tab[128] == [0, 0, 0, 1, 0, 0, 0, 0]
    assert tab[1] == [-1.54700e-04, 8.53777e-04, -2.76968e-03, 7.89295e-03, 9.98534e-01, -5.41054e-03, 1.24642e-03, -1.98993e-04]
    assert tab[2] == [-3.09412e-04, 1.70888e-03, -5.55134e-03, 1.58840e-02, 9.96891e-01, -1.07209e-02, 2.47942e-03, -3.96391e-04]
    assert tab[3] == [-4.64053e-04, 2.56486e-03, -8.34364e-03, 2.39714e-02, 9.95074e-01, -1.59305e-02, 3.69852e-03, -5.92100e-04]
    assert tab[64][:4] == [-6.77751e-03, 3.94578e-02, -1.42658e-01, 6.09836e-01] and tab[64][4:] == tab[64][3::-1]
    for r in range(129):
        assert tab[r] == tab[128 - r][::-1]
        assert abs(sum(tab[r]) - 1.0) < 2e-3


def test_fast_atan2f_accuracy():
    rng = np.random.default_rng(1)
    for _ in range(2000):
        y, x = rng.normal(size=2)
        assert abs(orc.fast_atan2f(float(np.float32(y)), float(np.float32(x))) - np.arctan2(np.float32(y), np.float32(x))) < 2e-5
    assert orc.fast_atan2f(0.0, 0.0) == 0.0
    assert abs(orc.fast_atan2f(1.0, 0.0) - np.pi / 2) < 1e-6


def test_fxpt_nco_sincos():
    # [GR] gr::fxpt: a 1024-segment piecewise-linear sine on a 32-bit angle; the table's recipe
    # (tools/gen_tables.py) reproduces the line upstream's sine_table.h opens with
    txt = open(os.path.join(os.path.dirname(__file__), "..", "oracle", "orc_tables.h")).read()
    assert "{  2.925817799165007e-09f,  7.219194364267018e-09f }," in txt
    worst = 0.0
    for ph in np.linspace(-3.1415925, 3.1415925, 20001):
        p = np.float32(ph)
        s, c = orc.nco_sincos(float(p))
        worst = max(worst, abs(s - np.sin(np.float64(p))), abs(c - np.cos(np.float64(p))))
    assert worst < 3.0e-6  # (chord error of a 2 pi / 1024 segment, halved by the offset: 2.4e-6)
    assert orc.lib().orc_fxpt_float_to_fixed(0.0) == 0
    assert orc.lib().orc_fxpt_float_to_fixed(float(np.float32(-np.pi))) == -2 ** 31
    assert orc.lib().orc_fxpt_float_to_fixed(float(np.float32(np.pi / 2))) == 2 ** 30
    assert orc.nco_sincos(0.0) == (float(np.float32(7.219194364267018e-09)), orc.nco_sincos(0.0)[1])
    assert abs(orc.nco_sincos(0.0)[1] - 1.0) < 3e-6


def test_division_by_pi_is_exact():
    # the kernels form float_to_fixed's x * 2^31 / PI without an IEEE division (aisx_common.h:
    # q = y * R, r = fma(-q, PI, y), q + r * R).  Every float in 1e-30 .. 1e30 was checked once
    # against the division; this keeps a dense sample: 2^21 consecutive floats up to PI, the
    # neighbourhoods of the fold, and phases far outside -PI .. PI
    import ctypes as C

    import emul_py as emu

    top = np.float32(np.pi).view(np.uint32)
    x = (top - np.arange(1 << 21, dtype=np.uint32)).view(np.float32)
    rng = np.random.default_rng(77)
    x = np.concatenate([x, -x, rng.uniform(-np.pi, np.pi, 1 << 20).astype(np.float32),
                        rng.normal(scale=40.0, size=1 << 16).astype(np.float32),
                        (rng.uniform(-1, 1, 1 << 16) * 10.0 ** rng.uniform(-30, 0, 1 << 16)).astype(np.float32),
                        np.array([0.0, -0.0, np.pi, -np.pi, 2 * np.pi, 1e-38, -1e-38], np.float32)])
    x = np.ascontiguousarray(x)
    a = np.zeros(x.size, np.int32)
    b = np.zeros(x.size, np.int32)
    vp = C.c_void_p
    emu.lib().emu_fxpt_float_to_fixed_n(vp(x.ctypes.data), vp(a.ctypes.data), C.c_long(x.size))
    orc.lib().orc_fxpt_float_to_fixed_n(vp(x.ctypes.data), vp(b.ctypes.data), C.c_long(x.size))
    assert np.array_equal(a, b)


def test_gmsk_template_shape():
    t = orc.gmsk_modulate_vector(4, 0.4, [1, 1, 0, 0] * 7)
    assert t.size == 224 * 4  # packed bytes -> 224 symbols
    assert np.allclose(np.abs(t), 1.0, atol=1e-6)


def test_select_form_of_fast_atan2f_equals_the_ladder():
    # the HIP kernels evaluate fast_atan2f with selects; the CPU lane model
    # exports that very function -- it must equal the oracle's if/else ladder bit for bit
    import ctypes as C

    import emul_py as emu

    L = emu.lib()
    L.emu_fast_atan2f.restype = C.c_float
    L.emu_fast_atan2f.argtypes = [C.c_float, C.c_float]
    rng = np.random.default_rng(8)
    vals = list(rng.normal(size=4000).astype(np.float32)) + [0.0, -0.0, 1.0, -1.0, 1e-38, -1e-38, 1e-45, -1e-45, 3e38,
                                                              -3e38, np.float32(np.inf), np.float32(-np.inf), np.float32(np.nan)]
    vals = [float(v) for v in vals]
    pairs = [(vals[i], vals[(7 * i + 3) % len(vals)]) for i in range(len(vals))]
    pairs += [(a, b) for a in vals[-13:] for b in vals[-13:]]
    for y, x in pairs:
        a = np.float32(L.emu_fast_atan2f(y, x))
        b = np.float32(orc.fast_atan2f(y, x))
        assert a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b)), (y, x, a, b)


# ---------------------------------------------------------------------------------------------
# Structural self-checks of the regenerated GNU Radio tables (tools/gen_tables.py writes the same
# numbers into oracle/orc_tables.h and gr-ais_amd/csrc/aisx_tables.h).  Not a pin -- the tables are
# regenerated from upstream's published recipes, upstream's files are not in the image -- but
# properties any correct copy of those tables has, checked on BOTH generated headers.
# ---------------------------------------------------------------------------------------------
def _parse_table(path, name):
    import re

    txt = open(path).read()
    m = re.search(r"%s\s*\[[^=]*=\s*\{(.*?)\};" % name, txt, flags=re.S)
    assert m, (path, name)
    body = re.sub(r"/\*.*?\*/|//[^\n]*", "", m.group(1), flags=re.S)
    return np.array([float(v.rstrip("fF")) for v in re.findall(r"[-+]?\d[\d.]*(?:[eE][-+]?\d+)?[fF]?", body)], dtype=np.float64)


@pytest.mark.parametrize("path,prefix", [("oracle/orc_tables.h", "orc"), ("gr-ais_amd/csrc/aisx_tables.h", "aisx")])
def test_generated_tables_have_the_structure_of_upstreams(path, prefix):
    full = os.path.join(ROOT, path)
    # mmse_fir_interpolator taps: 129 rows of 8; row r reversed == row 128 - r (the interpolator for
    # mu and for 1 - mu are mirror images); rows sum to ~1 (DC gain); rows 0 / 128 are pure delays;
    # the centre row is symmetric
    t = _parse_table(full, prefix + "_mmse_taps").reshape(129, 8)
    assert np.array_equal(t[::-1, ::-1], t)
    assert np.all(np.abs(t.sum(axis=1) - 1.0) < 2e-3)
    assert t[0].tolist() == [0, 0, 0, 0, 1, 0, 0, 0] and t[128].tolist() == [0, 0, 0, 1, 0, 0, 0, 0]
    assert np.array_equal(t[64], t[64][::-1])
    # every tap is printed to six significant digits, as upstream's %12.5e file is
    assert all(float("%.5e" % v) == v for v in t.ravel())
    # the main taps move monotonically with mu
    assert np.all(np.diff(t[:, 4]) < 0) and np.all(np.diff(t[:, 3]) > 0)
    # fast_atan2f: 257 entries of atan(i / 255), last one repeated
    a = _parse_table(full, prefix + "_atan_table")
    assert a.size == 257 and a[0] == 0.0 and a[256] == a[255]
    assert np.max(np.abs(a[:256] - np.arctan(np.arange(256) / 255.0))) < 5e-8
    assert abs(a[255] - np.pi / 4) < 1e-7
    # gr::fxpt sine table: 1024 x {slope, offset}; the line of entry i evaluated at the start of
    # entry i + 1 continues into that entry's own line (continuity of the piecewise-linear sine),
    # and its value at the entry's first angle is sin of that angle to table accuracy
    st = _parse_table(full, prefix + "_sine_table").reshape(1024, 2)
    x0 = (np.arange(1024, dtype=np.int64) << 22).astype(np.uint32).astype(np.int32).astype(np.float64)  # signed 32-bit angle
    ux = (np.arange(1024, dtype=np.int64) << 22) >> 1  # fxpt::sin evaluates the line at (unsigned angle) >> 1
    val = st[:, 0] * ux + st[:, 1]
    assert np.max(np.abs(val - np.sin(x0 * np.pi / 2147483648.0))) < 3e-6  # (a min-max line fit: the chord error of a 2 pi / 1024 step, (pi/512)^2 / 16)
    end = st[:, 0] * (ux + (1 << 21)) + st[:, 1]  # the line of entry i at the first angle of entry i + 1
    assert np.max(np.abs(end[:-1] - val[1:])) < 6e-6 and abs(end[-1] - val[0]) < 6e-6
