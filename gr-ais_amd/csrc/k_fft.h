// k_fft.h -- register-resident radix-4/8/16 DFT butterflies (natural order in,
// natural order out; all indices compile-time so the arrays stay in VGPRs).
// Forward = e^{-j2pi nk/R}; INV = conjugate kernel (unnormalised).
//
// Every operation is one of five complex primitives of the execution context:
//   cadd(a, b) = a + b          csub(a, b) = a - b
//   add_mj(a, b) = a - j b      add_pj(a, b) = a + j b
//   cmul_sel<K, CS, CNEG, SS, SNEG>(a) = a (c + j s), c = +-K[CS], s = +-K[SS], K one of the two
//   constant pairs (cos(pi/8), sin(pi/8)) and (sqrt(1/2), sqrt(1/2))
// On gfx950 each is exactly one packed-fp32 instruction (two for cmul_k): the +-j rotations
// ride on the op_sel / neg modifiers of v_pk_add_f32 instead of costing register moves
// (aisx_devctx.h); the CPU lane model does the same arithmetic in plain C++.
#pragma once
#include "aisx_common.h"

namespace aisx {

template <bool INV, class Ctx>
AISX_HD void dft4(const Ctx& cx, cf& a0, cf& a1, cf& a2, cf& a3)
{
    const cf t0 = cx.cadd(a0, a2), t1 = cx.csub(a0, a2), t2 = cx.cadd(a1, a3), t3 = cx.csub(a1, a3);
    a0 = cx.cadd(t0, t2);
    a2 = cx.csub(t0, t2);
    if (!INV) { // X1 = t1 - j t3 ; X3 = t1 + j t3
        a1 = cx.add_mj(t1, t3);
        a3 = cx.add_pj(t1, t3);
    } else {
        a1 = cx.add_pj(t1, t3);
        a3 = cx.add_mj(t1, t3);
    }
}

// multiply by W16^m (forward) or its conjugate (INV), m compile-time
template <int M, bool INV, class Ctx>
AISX_HD cf mul_w16(const Ctx& cx, cf a)
{
    constexpr int m = M & 15;
    const cf z = mk(0.f, 0.f);
    if (m == 0)
        return a;
    if (m == 4) // -j (fwd) / +j (inv)
        return INV ? cx.add_pj(z, a) : cx.add_mj(z, a);
    if (m == 8)
        return cx.csub(z, a);
    if (m == 12)
        return INV ? cx.add_mj(z, a) : cx.add_pj(z, a);
    // general: a (c + j wi) with c = +-K[CS], wi = +-K[SS], K = (cos(pi/8), sin(pi/8)) or
    // (sqrt(1/2), sqrt(1/2)): the context gets the selection, not the numbers, so that on the
    // device the two constant pairs (scalar registers) serve every case through operand-select
    // and negate modifiers
    //   forward twiddle of index m: c = cos(2 pi m / 16), wi = -sin(2 pi m / 16); INV: wi = +sin
    constexpr int KR2 = 1, KC1 = 0;
    switch (m) {
    case 1: return cx.template cmul_sel<KC1, 0, false, 1, !INV>(a);            //  C1, -+S1
    case 2: return cx.template cmul_sel<KR2, 0, false, 1, !INV>(a);            //  R2, -+R2
    case 3: return cx.template cmul_sel<KC1, 1, false, 0, !INV>(a);            //  S1, -+C1
    case 5: return cx.template cmul_sel<KC1, 1, true, 0, !INV>(a);             // -S1, -+C1
    case 6: return cx.template cmul_sel<KR2, 0, true, 1, !INV>(a);             // -R2, -+R2
    case 7: return cx.template cmul_sel<KC1, 0, true, 1, !INV>(a);             // -C1, -+S1
    case 9: return cx.template cmul_sel<KC1, 0, true, 1, INV>(a);              // -C1, +-S1
    case 10: return cx.template cmul_sel<KR2, 0, true, 1, INV>(a);             // -R2, +-R2
    case 11: return cx.template cmul_sel<KC1, 1, true, 0, INV>(a);             // -S1, +-C1
    case 13: return cx.template cmul_sel<KC1, 1, false, 0, INV>(a);            //  S1, +-C1
    case 14: return cx.template cmul_sel<KR2, 0, false, 1, INV>(a);            //  R2, +-R2
    default: return cx.template cmul_sel<KC1, 0, false, 1, INV>(a);            //  C1, +-S1  (15)
    }
}

// 8-point DFT, n = 2*n1 + n2 (N1 = 4, N2 = 2), k = k1 + 4*k2
template <bool INV, class Ctx>
AISX_HD void dft8(const Ctx& cx, cf (&x)[8])
{
    dft4<INV>(cx, x[0], x[2], x[4], x[6]); // n2 = 0 : y[k1][0] at x[2*k1]
    dft4<INV>(cx, x[1], x[3], x[5], x[7]); // n2 = 1 : y[k1][1] at x[2*k1+1]
    x[3] = mul_w16<2, INV>(cx, x[3]);      // W8^1
    // W8^2 = -+j rides on the butterfly below
    x[7] = mul_w16<6, INV>(cx, x[7]);      // W8^3
    cf o[8];
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) {
        if (k1 == 2) {
            o[k1] = INV ? cx.add_pj(x[4], x[5]) : cx.add_mj(x[4], x[5]);
            o[k1 + 4] = INV ? cx.add_mj(x[4], x[5]) : cx.add_pj(x[4], x[5]);
        } else {
            o[k1] = cx.cadd(x[2 * k1], x[2 * k1 + 1]);
            o[k1 + 4] = cx.csub(x[2 * k1], x[2 * k1 + 1]);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++)
        x[k] = o[k];
}

// 16-point DFT, n = 4*n1 + n2, k = k1 + 4*k2
template <bool INV, class Ctx>
AISX_HD void dft16(const Ctx& cx, cf (&x)[16])
{
    dft4<INV>(cx, x[0], x[4], x[8], x[12]);
    dft4<INV>(cx, x[1], x[5], x[9], x[13]);
    dft4<INV>(cx, x[2], x[6], x[10], x[14]);
    dft4<INV>(cx, x[3], x[7], x[11], x[15]);
    // y[k1][n2] at x[4*k1 + n2]; twiddle W16^{n2*k1}
    x[5] = mul_w16<1, INV>(cx, x[5]);
    x[6] = mul_w16<2, INV>(cx, x[6]);
    x[7] = mul_w16<3, INV>(cx, x[7]);
    x[9] = mul_w16<2, INV>(cx, x[9]);
    x[10] = mul_w16<4, INV>(cx, x[10]);
    x[11] = mul_w16<6, INV>(cx, x[11]);
    x[13] = mul_w16<3, INV>(cx, x[13]);
    x[14] = mul_w16<6, INV>(cx, x[14]);
    x[15] = mul_w16<9, INV>(cx, x[15]);
    dft4<INV>(cx, x[0], x[1], x[2], x[3]);
    dft4<INV>(cx, x[4], x[5], x[6], x[7]);
    dft4<INV>(cx, x[8], x[9], x[10], x[11]);
    dft4<INV>(cx, x[12], x[13], x[14], x[15]);
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
