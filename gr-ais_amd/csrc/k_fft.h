// k_fft.h -- register-resident radix-4/8/16 DFT butterflies (natural order in,
// natural order out; all indices compile-time so the arrays stay in VGPRs).
// Forward = e^{-j2pi nk/R}; INV = conjugate kernel (unnormalised).
#pragma once
#include "aisx_common.h"

namespace aisx {

template <bool INV>
AISX_HD void dft4(cf& a0, cf& a1, cf& a2, cf& a3)
{
    cf t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = a1 - a3;
    a0 = t0 + t2;
    a2 = t0 - t2;
    if (!INV) { // X1 = t1 - j t3 ; X3 = t1 + j t3
        a1 = mk(t1.re + t3.im, t1.im - t3.re);
        a3 = mk(t1.re - t3.im, t1.im + t3.re);
    } else {
        a1 = mk(t1.re - t3.im, t1.im + t3.re);
        a3 = mk(t1.re + t3.im, t1.im - t3.re);
    }
}

// multiply by W16^m (forward) or its conjugate (INV), m compile-time
template <int M, bool INV>
AISX_HD cf mul_w16(cf a)
{
    constexpr float C1 = 0.92387953251128673848f; // cos(pi/8)
    constexpr float S1 = 0.38268343236508978178f; // sin(pi/8)
    constexpr float R2 = 0.70710678118654752440f;
    constexpr int m = M & 15;
    if (m == 0)
        return a;
    if (m == 4) // -j (fwd) / +j (inv)
        return INV ? mk(-a.im, a.re) : mk(a.im, -a.re);
    if (m == 8)
        return mk(-a.re, -a.im);
    if (m == 12)
        return INV ? mk(a.im, -a.re) : mk(-a.im, a.re);
    // general: w = (c, -s) forward
    float c = 0.f, s = 0.f;
    switch (m) {
    case 1: c = C1; s = S1; break;
    case 2: c = R2; s = R2; break;
    case 3: c = S1; s = C1; break;
    case 5: c = -S1; s = C1; break;
    case 6: c = -R2; s = R2; break;
    case 7: c = -C1; s = S1; break;
    case 9: c = -C1; s = -S1; break;
    case 10: c = -R2; s = -R2; break;
    case 11: c = -S1; s = -C1; break;
    case 13: c = S1; s = -C1; break;
    case 14: c = R2; s = -R2; break;
    default: c = C1; s = -S1; break; // 15
    }
    float wi = INV ? s : -s;
    // (a.re + j a.im)(c + j wi)
    return mk(fmaf(-a.im, wi, a.re * c), fmaf(a.im, c, a.re * wi));
}

// 8-point DFT, n = 2*n1 + n2 (N1 = 4, N2 = 2), k = k1 + 4*k2
template <bool INV>
AISX_HD void dft8(cf (&x)[8])
{
    dft4<INV>(x[0], x[2], x[4], x[6]); // n2 = 0 : y[k1][0] at x[2*k1]
    dft4<INV>(x[1], x[3], x[5], x[7]); // n2 = 1 : y[k1][1] at x[2*k1+1]
    x[3] = mul_w16<2, INV>(x[3]);      // W8^1
    x[5] = mul_w16<4, INV>(x[5]);      // W8^2
    x[7] = mul_w16<6, INV>(x[7]);      // W8^3
    cf o[8];
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) {
        o[k1] = x[2 * k1] + x[2 * k1 + 1];
        o[k1 + 4] = x[2 * k1] - x[2 * k1 + 1];
    }
#pragma unroll
    for (int k = 0; k < 8; k++)
        x[k] = o[k];
}

// 16-point DFT, n = 4*n1 + n2, k = k1 + 4*k2
template <bool INV>
AISX_HD void dft16(cf (&x)[16])
{
    dft4<INV>(x[0], x[4], x[8], x[12]);
    dft4<INV>(x[1], x[5], x[9], x[13]);
    dft4<INV>(x[2], x[6], x[10], x[14]);
    dft4<INV>(x[3], x[7], x[11], x[15]);
    // y[k1][n2] at x[4*k1 + n2]; twiddle W16^{n2*k1}
    x[5] = mul_w16<1, INV>(x[5]);
    x[6] = mul_w16<2, INV>(x[6]);
    x[7] = mul_w16<3, INV>(x[7]);
    x[9] = mul_w16<2, INV>(x[9]);
    x[10] = mul_w16<4, INV>(x[10]);
    x[11] = mul_w16<6, INV>(x[11]);
    x[13] = mul_w16<3, INV>(x[13]);
    x[14] = mul_w16<6, INV>(x[14]);
    x[15] = mul_w16<9, INV>(x[15]);
    dft4<INV>(x[0], x[1], x[2], x[3]);
    dft4<INV>(x[4], x[5], x[6], x[7]);
    dft4<INV>(x[8], x[9], x[10], x[11]);
    dft4<INV>(x[12], x[13], x[14], x[15]);
    // x[4*k1 + k2] = X[k1 + 4*k2] -> transpose to natural order
    cf o[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++)
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++)
            o[k1 + 4 * k2] = x[4 * k1 + k2];
#pragma unroll
    for (int k = 0; k < 16; k++)
        x[k] = o[k];
}

} // namespace aisx
