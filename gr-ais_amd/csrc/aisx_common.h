// aisx_common.h -- shared scalar/complex helpers for the HIP kernels.
//
// Kernel BODIES are written as templates over an execution context `Ctx`
// (thread id, LDS base, barrier, wave ballot/shuffle, global atomics).  The
// product instantiates them with DevCtx (aisx_devctx.h, real gfx950 wave64
// builtins); tests/emul instantiates the same bodies with a thread-per-lane
// CPU model so the index arithmetic can be checked against the oracle without a
// GPU.  Everything is compiled with -ffp-contract=off and uses explicit fmaf()
// where a fused multiply-add is wanted, so both instantiations round alike.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define AISX_DI __device__ __forceinline__
#define AISX_HD __host__ __device__ __forceinline__
#else
#define AISX_DI inline
#define AISX_HD inline
#endif

namespace aisx {

struct cf {
    float re, im;
};

AISX_HD cf mk(float re, float im)
{
    cf r;
    r.re = re;
    r.im = im;
    return r;
}
AISX_HD cf operator+(cf a, cf b) { return mk(a.re + b.re, a.im + b.im); }
AISX_HD cf operator-(cf a, cf b) { return mk(a.re - b.re, a.im - b.im); }
AISX_HD cf cconj(cf a) { return mk(a.re, -a.im); }

// LDS / global accesses whose alignment the compiler cannot see for itself (pointers that
// alternate between two images): one item as a 64-bit access, two neighbours as a 128-bit one.
// The caller guarantees the alignment (8 resp. 16 bytes).
struct alignas(8) cf_a8 { float re, im; };
struct alignas(16) cf_a16 { float re0, im0, re1, im1; };
AISX_HD cf ld8(const cf* p)
{
    const cf_a8 v = *reinterpret_cast<const cf_a8*>(p);
    return mk(v.re, v.im);
}
AISX_HD void st8(cf* p, cf v)
{
    cf_a8 t;
    t.re = v.re;
    t.im = v.im;
    *reinterpret_cast<cf_a8*>(p) = t;
}
AISX_HD void ld16(const cf* p, cf& a, cf& b)
{
    const cf_a16 v = *reinterpret_cast<const cf_a16*>(p);
    a = mk(v.re0, v.im0);
    b = mk(v.re1, v.im1);
}
AISX_HD void st16(cf* p, cf a, cf b)
{
    cf_a16 t;
    t.re0 = a.re;
    t.im0 = a.im;
    t.re1 = b.re;
    t.im1 = b.im;
    *reinterpret_cast<cf_a16*>(p) = t;
}

// std::complex<float> product as libstdc++ evaluates it: (ac - bd, ad + bc),
// every operation rounded to float, no fusing.  Used on the bit-exact paths
// (timing recovery, NCO mix, AGC).
AISX_HD cf cmul_exact(cf a, cf b) { return mk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }

// Fused complex product for the FFT correlator (tolerance path): 2 mul + 2 fma.
AISX_HD cf cmul_fma(cf a, cf b)
{
    return mk(fmaf(-a.im, b.im, a.re * b.re), fmaf(a.im, b.re, a.re * b.im));
}
AISX_HD cf cmul_conj_fma(cf a, cf b) // a * conj(b)
{
    return mk(fmaf(a.im, b.im, a.re * b.re), fmaf(a.im, b.re, -(a.re * b.im)));
}

// volk_32fc_magnitude_squared_32f (generic kernel): re*re + im*im, unfused.
AISX_HD float mag2(cf a) { return a.re * a.re + a.im * a.im; }

// float division, correctly rounded on both sides
AISX_HD float fdiv_rn(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __fdiv_rn(a, b);
#else
    return a / b;
#endif
}

// gr::branchless_clip (gnuradio/math.h)
AISX_HD float branchless_clip(float x, float clip)
{
    float x1 = fabsf(x + clip);
    float x2 = fabsf(x - clip);
    x1 -= x2;
    return 0.5f * x1;
}

AISX_HD int aisx_popc64(unsigned long long v) { return __builtin_popcountll(v); }

// gr::fast_atan2f (gnuradio-runtime fast_atan2f.cc); `tab` = 257-entry table.
// Written with selects instead of the upstream if/else ladder (a divergent
// ladder costs a wave every arm); every arithmetic operation and operand is the
// one the upstream code executes on the arm it would take, so the result is
// bit-identical (tests/test_oracle_kat.py checks it against the ladder).
AISX_HD float fast_atan2f_tab(float y, float x, const float* tab)
{
    const float TAN_MAP_RES = 0.003921569f;
    const float PI_F = 3.14159265358979323846f, PI2_F = 1.57079632679489661923f;
    const float y_abs = fabsf(y), x_abs = fabsf(x);
    const bool ylt = y_abs < x_abs;
    const float z = fdiv_rn(ylt ? y_abs : x_abs, ylt ? x_abs : y_abs);
    float alpha = z * 255.0f;
    const int index = ((int)alpha) & 0xff;
    alpha -= (float)index;
    const float t0 = tab[index], t1 = tab[index + 1];
    float base_angle = t0 + (t1 - t0) * alpha;
    base_angle = (z < TAN_MAP_RES) ? z : base_angle;
    const bool xge = x >= 0.0f, yge = y >= 0.0f;
    // x_abs > y_abs: -45..45 or 135..225
    const float a_h = xge ? (yge ? base_angle : -base_angle) : (yge ? (PI_F - base_angle) : (base_angle - PI_F));
    // otherwise: 45..135 or -135..-45
    const float a_v = yge ? (xge ? (PI2_F - base_angle) : (PI2_F + base_angle))
                          : (xge ? (-PI2_F + base_angle) : (-PI2_F - base_angle));
    const float angle = (x_abs > y_abs) ? a_h : a_v;
    return ((y_abs > 0.0f) || (x_abs > 0.0f)) ? angle : 0.0f;
}

// [GR] gr::fxpt (gnuradio-runtime fxpt.h, 3.7 / 3.8): the fixed-point sin / cos behind
// frequency_modulator_fc (angle = float_to_fixed(d_phase); sincos(angle, &oq, &oi)).  A 32-bit
// angle (2^31 = pi); its top 10 bits pick a {slope, offset} pair of s_sine_table (`tab`, 1024 x 2
// floats: aisx_tables.h, staged in LDS by the kernels), the line is evaluated at ux >> 1 in float,
// multiply and add rounded separately.  Integer and fp32 work only: both sides agree bit for bit.
AISX_HD int fxpt_float_to_fixed(float x)
{
    const float PI = 3.14159265358979323846f, TAU = 2.0f * 3.14159265358979323846f, TWO_TO_THE_31 = 2147483648.0f;
    // Fold x into -PI .. PI: d = (int)floor(x / TAU + 0.5); x -= d * TAU.  For -PI <= x < PI the
    // quotient lies in [-0.5, 0.5), the sum (formed in double, nothing is rounded) in [0, 1), d is 0
    // and the fold changes nothing -- which is every phase frequency_modulator_fc's own wrap
    // produces: one compare instead of a division and three double-precision operations.
    if (!(x >= -PI && x < PI)) {
        const int d = (int)floor((double)fdiv_rn(x, TAU) + 0.5);
        x -= (float)d * TAU;
    }
    // (int)(x * 2^31 / PI).  The float division by the constant PI is formed as Markstein's
    // q = y * R; r = fma(-q, PI, y); q + r * R with R = fl(1 / PI): three instructions instead of
    // the eleven of an IEEE division, and the same float for EVERY y in 1e-30 .. 1e30 (checked
    // exhaustively, tests/test_oracle_kat.py::test_division_by_pi_is_exact holds the sampled form).
    const float RCP_PI = 0.318309886183790671538f;
    const float y = x * TWO_TO_THE_31;
    const float q = y * RCP_PI;
    return (int)fmaf(fmaf(-q, PI, y), RCP_PI, q);
}
AISX_HD void nco_sincos(float phase, const float* tab, float* s, float* c)
{
    const cf* T = reinterpret_cast<const cf*>(tab); // {slope, offset} pairs, 8-byte aligned (LDS or aisx_tables.h)
    const unsigned x = (unsigned)fxpt_float_to_fixed(phase);
    const unsigned xc = x + 0x40000000u;
    const cf es = ld8(T + (x >> 22)), ec = ld8(T + (xc >> 22));
    *s = es.re * (float)(x >> 1) + es.im;
    *c = ec.re * (float)(xc >> 1) + ec.im;
}

// [GR] frequency_modulator_fc: d_phase = fmod(d_phase + pi, 2 pi) - pi.  fmod is an
// exact operation; for |u| < 4 pi it is u or u -/+ 2 pi (exact by Sterbenz).
AISX_HD float nco_wrap(float ph)
{
    const float F_PI = 3.14159265358979323846f;
    const float TWO_PI = 2.0f * F_PI;
    const float u = ph + F_PI;
    float r;
    const float au = fabsf(u);
    if (au < TWO_PI)
        r = u;
    else if (au < 2.0f * TWO_PI)
        r = (u > 0.f) ? (u - TWO_PI) : (u + TWO_PI);
    else
        r = fmodf(u, TWO_PI);
    return r - F_PI;
}

// the same for |ph + pi| < 4 pi: u, or u -/+ 2 pi -- as selects
AISX_HD float nco_wrap_small(float ph)
{
    const float F_PI = 3.14159265358979323846f;
    const float TWO_PI = 2.0f * F_PI;
    const float u = ph + F_PI;
    const float w = u - copysignf(TWO_PI, u);
    const float r = (fabsf(u) < TWO_PI) ? u : w;
    return r - F_PI;
}

// The NCO phase walk (k_freqsync.h fs_walk_body) leaves every NCO_CK-th phase in memory; readers (k_agc.h) walk
// the ones in between again.  2, 4 or 8: bytes per sample against the length of the dependent chains.
#ifndef NCO_CKN
#define NCO_CKN 8
#endif
constexpr int NCO_CK = NCO_CKN;
static_assert(NCO_CK == 2 || NCO_CK == 4 || NCO_CK == 8, "checkpoints every 2, 4 or 8 items");
// one step of the recurrence; `small`: |d| < 6 (then the fmod of the wrap is a select: the same value)
AISX_HD float nco_phase_step(float ph, float d, bool small) { return small ? nco_wrap_small(ph + d) : nco_wrap(ph + d); }

constexpr int NCO_TAB_FLOATS = 2048; // s_sine_table: 1024 x {slope, offset}

// std::abs(std::complex<float>) as glibc's hypotf evaluates it
AISX_HD float cabs_f(cf a) { return (float)sqrt((double)a.re * (double)a.re + (double)a.im * (double)a.im); }

// ------------------------------------------------------------------
// Tag record shared by the C-ABI (must match aisx_tag in include/aisx.h)
// ------------------------------------------------------------------
struct tag_rec {
    uint64_t offset;
    double value;
    int32_t key;
    int32_t chan;
};
enum { KEY_CORR_START = 0, KEY_PHASE_EST = 1, KEY_TIME_EST = 2, KEY_CORR_EST = 3 };

} // namespace aisx
