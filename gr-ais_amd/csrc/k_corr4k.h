// k_corr4k.h -- the F = 4096 build of the corr_est_cc correlator (same contract as
// corr_main_body / corr_inith_body in k_corr.h; used for templates longer than 512
// samples: with L = F - N valid outputs per tile it spends 27 % fewer flops per sample
// than F = 2048 at N = 896, and it is the only build for 1024 < N <= 2048).
//
#pragma once
#include "k_corr.h"

namespace aisx {

// FFT: F = 4096 = 16 x 16 x 16, 256 threads (4 waves) per transform, 16 points
// per thread in VGPRs, three radix-16 register passes per direction with two LDS
// exchanges.  Forward is DIF (natural in, digit-reversed positions out), the
// template spectrum H is produced by the same forward pass and so already sits
// in those positions, inverse is DIT (natural out): no reordering pass.  With
// L = F - N valid outputs per tile the fixed per-tile latencies (global load,
// barriers, twiddle/H fetch) are spread over 3200 samples at N = 896.
// LDS image: 16 rows x 272 complex (row pad 16 => the stride-16 radix-16 pass
// touches 64 distinct banks per half wave); inside each 16-element group the
// 16-byte chunk index is XOR-ed with (k2 >> 1) so the stride-1 pass reads
// ds_read_b128 conflict free.

constexpr int CF4_F = 4096;    // FFT size
constexpr int CF4_T = 256;     // threads per transform
constexpr int CF4_ROW = 272;   // LDS row pitch in complex elements
constexpr int CF4_LDS_ELEMS = 16 * CF4_ROW + 256; // data + W_256^{k2*n3} table
constexpr int CF4_LDS_BYTES = CF4_LDS_ELEMS * 8;

// position of (row k1|n1, column col = k2*16 + n3) in the LDS image
AISX_HD int cf4_pos(int row, int col)
{
    const int k2 = col >> 4, n3 = col & 15;
    return row * CF4_ROW + (k2 << 4) + ((((n3 >> 1) ^ (k2 >> 1)) << 1) | (n3 & 1));
}


// x[k] *= w^k (conjugated if INV), k = 1..15, from the stored powers w, w^2, w^4, w^8:
// at most three extra products per twiddle, no table traffic inside the tile loop
template <bool INV>
AISX_DI void cf4_pow_twiddles(cf (&x)[16], const cf (&wp)[4])
{
    const cf w1 = wp[0], w2 = wp[1], w4 = wp[2], w8 = wp[3];
    const cf w3 = cmul_fma(w1, w2), w5 = cmul_fma(w4, w1), w6 = cmul_fma(w4, w2);
    const cf w7 = cmul_fma(w4, w3);
#define AISX_AP(k, w) x[k] = INV ? cmul_conj_fma(x[k], (w)) : cmul_fma(x[k], (w))
    AISX_AP(1, w1);
    AISX_AP(2, w2);
    AISX_AP(3, w3);
    AISX_AP(4, w4);
    AISX_AP(5, w5);
    AISX_AP(6, w6);
    AISX_AP(7, w7);
    AISX_AP(8, w8);
    AISX_AP(9, cmul_fma(w8, w1));
    AISX_AP(10, cmul_fma(w8, w2));
    AISX_AP(11, cmul_fma(w8, w3));
    AISX_AP(12, cmul_fma(w8, w4));
    AISX_AP(13, cmul_fma(w8, w5));
    AISX_AP(14, cmul_fma(w8, w6));
    AISX_AP(15, cmul_fma(w8, w7));
#undef AISX_AP
}

// ---- forward passes shared by init and main -------------------------------
// On entry x[n1] = w[t + 256*n1].  On exit x[k3] holds the spectrum at position
// (q = t, k3), i.e. logical index t*16 + k3 (frequency k1 + 16*k2 + 256*k3 with
// k1 = t >> 4, k2 = t & 15).
template <class Ctx>
AISX_DI void cf4_forward(Ctx& cx, cf (&x)[16], const cf (&wp)[4], cf* ldsX, const cf* ldsT)
{
    const int t = cx.tid();
    dft16<false>(cx, x);
    cf4_pow_twiddles<false>(x, wp); // W_4096^{k1*t}
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++)
        ldsX[cf4_pos(k1, t)] = x[k1];
    cx.sync();
    {
        const int k1 = t >> 4, n3 = t & 15;
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++)
            x[n2] = ldsX[cf4_pos(k1, n2 * 16 + n3)];
        dft16<false>(cx, x);
#pragma unroll
        for (int k2 = 1; k2 < 16; k2++)
            x[k2] = cmul_fma(x[k2], ldsT[k2 * 16 + n3]); // W_256^{k2*n3}
#pragma unroll
        for (int k2 = 0; k2 < 16; k2++)
            ldsX[cf4_pos(k1, k2 * 16 + n3)] = x[k2];
    }
    cx.sync();
    {
        const int k1 = t >> 4, k2 = t & 15, swz = k2 >> 1;
        const int base = k1 * CF4_ROW + k2 * 16;
#pragma unroll
        for (int pr = 0; pr < 8; pr++) {
            const int ch = base + 2 * (pr ^ swz);
            x[2 * pr] = ldsX[ch];
            x[2 * pr + 1] = ldsX[ch + 1];
        }
        dft16<false>(cx, x);
    }
}


template <class Ctx>
AISX_DI void cf4_setup(Ctx& cx, const cf* wtab, cf* ldsT, cf (&wp)[4])
{
    const int t = cx.tid();
    wp[0] = wtab[t];
    wp[1] = wtab[2 * t];
    wp[2] = wtab[4 * t];
    wp[3] = wtab[8 * t];
    ldsT[t] = wtab[(16 * (t >> 4) * (t & 15)) & (CF4_F - 1)]; // W_256^{k2*n3}, index k2*16+n3
    cx.sync();
}

template <class Ctx>
AISX_DI void corr4_inith_body(Ctx& cx, const CorrInitParams& p)
{
    const int t = cx.tid();
    cf* lds = (cf*)cx.lds();
    cf* ldsX = lds;
    cf* ldsT = lds + 16 * CF4_ROW;
    cf wp[4];
    cf4_setup(cx, p.wtab, ldsT, wp);
    cf x[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; n1++)
        x[n1] = p.taps_scaled[t + CF4_T * n1];
    cf4_forward(cx, x, wp, ldsX, ldsT);
#pragma unroll
    for (int k3 = 0; k3 < 16; k3++)
        p.Hpos[t * 16 + k3] = x[k3];
}

template <class Ctx>
AISX_DI void corr4_main_body(Ctx& cx, const CorrParams& p)
{
    const int t = cx.tid();
    const int c = cx.by();
    const int seg = cx.bx();
    cf* lds = (cf*)cx.lds();
    cf* ldsX = lds;
    cf* ldsT = lds + 16 * CF4_ROW;

    const int N = p.N, L = p.L, n = p.n;
    const cf* xin = p.in + (long)c * p.in_stride;
    cf* xout = p.out + (long)c * p.out_stride;
    cf* xcorr = p.corr + (long)c * p.corr_stride;
    const cf* hist = p.hist_in + (long)c * N;
    unsigned long long* abits = p.abits + (long)c * p.abits_stride;

    cf wp[4];
    cf4_setup(cx, p.wtab, ldsT, wp);
    // value n1 of this thread is window item i = t + 256 n1; in an interior tile it is a
    // correlation output iff i >= N (i < F = N + L always)
    unsigned vmask_int = 0;
#pragma unroll
    for (int n1 = 0; n1 < 16; n1++)
        if (t + CF4_T * n1 >= N)
            vmask_int |= 1u << n1;

    for (int tile = 0; tile < p.tiles_per_seg; tile++) {
        const int k0 = (seg * p.tiles_per_seg + tile) * L;
        if (k0 >= n)
            break;
        cf x[16];
        // a tile whose whole window and all L outputs lie inside this call's items (all but the
        // first and the last of a channel): no per-element bounds tests, the few that remain
        // (i < L, i >= N) are decided per 256-element slice in scalar code
        const bool interior = (k0 - N >= 0) && (k0 + L <= n);
        if (interior) {
            const cf* w = xin + (k0 - N);
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++)
                x[n1] = (w + CF4_T * n1)[(unsigned)t]; // uniform base + 32-bit lane offset
            // A2: out[k0 + i] = stream[k0 + i - N] = w[i]   (lib/corr_est_cc_impl.cc:184)
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                const int lo = CF4_T * n1;
                cf* on = xout + (k0 + lo);
                if (lo + CF4_T <= L)
                    on[(unsigned)t] = x[n1];
                else if (lo < L && lo + t < L)
                    on[(unsigned)t] = x[n1];
            }
        } else {
            // window w[i] = stream[k0 - N + i]; stream index < 0 comes from the history
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                const int i = t + CF4_T * n1;
                const int s = k0 - N + i;
                cf val = mk(0.f, 0.f);
                if (s < 0)
                    val = hist[N + s];
                else if (s < n)
                    val = xin[s];
                x[n1] = val;
            }
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                const int i = t + CF4_T * n1;
                if (i < L && k0 + i < n)
                    xout[k0 + i] = x[n1];
            }
            if (p.corr_hist_zero) {
#pragma unroll
                for (int n1 = 0; n1 < 16; n1++)
                    if (k0 - N + t + CF4_T * n1 < 0)
                        x[n1] = mk(0.f, 0.f);
            }
        }
        cf4_forward(cx, x, wp, ldsX, ldsT);
        // spectrum x H (position order, 128 contiguous bytes per thread), inverse radix-16
        {
            const int k1 = t >> 4, k2 = t & 15, swz = k2 >> 1;
            const int base = k1 * CF4_ROW + k2 * 16;
            const cf* Hq = p.Hpos + t * 16;
#pragma unroll
            for (int k3 = 0; k3 < 16; k3++)
                x[k3] = cmul_fma(x[k3], Hq[k3]);
            dft16<true>(cx, x);
#pragma unroll
            for (int pr = 0; pr < 8; pr++) {
                const int ch = base + 2 * (pr ^ swz);
                ldsX[ch] = x[2 * pr];
                ldsX[ch + 1] = x[2 * pr + 1];
            }
        }
        cx.sync();
        {
            const int k1 = t >> 4, n3 = t & 15;
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) {
                cf a = ldsX[cf4_pos(k1, k2 * 16 + n3)];
                x[k2] = (k2 == 0) ? a : cmul_conj_fma(a, ldsT[k2 * 16 + n3]);
            }
            dft16<true>(cx, x);
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++)
                ldsX[cf4_pos(k1, n2 * 16 + n3)] = x[n2];
        }
        cx.sync();
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++)
            x[k1] = ldsX[cf4_pos(k1, t)];
        cf4_pow_twiddles<true>(x, wp);
        dft16<true>(cx, x);
        // y[i] = corr[k0 + i - N]; A4 mag^2 (:191) and the threshold test (:197).  Which of the
        // thread's 16 values are correlation outputs is a 16-bit mask (constant over interior
        // tiles); the threshold test adds to a hit mask without branching; only a wave with a
        // hit (rare) walks its bits.
        unsigned vmask = vmask_int;
        if (!interior) {
            vmask = 0;
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                const int m = t + CF4_T * n1 - N;
                if (m >= 0 && m < L && k0 + m < n)
                    vmask |= 1u << n1;
            }
        }
        const int kb = k0 + t - N; // output index of value n1: kb + 256 n1
        if (p.dense_corr) {
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++)
                if ((vmask >> n1) & 1u)
                    xcorr[kb + CF4_T * n1] = x[n1];
        }
        unsigned hit = 0;
#pragma unroll
        for (int n1 = 0; n1 < 16; n1++) {
            const float mg = mag2(x[n1]);
            hit |= (!(mg <= p.thresh)) ? (1u << n1) : 0u;
        }
        hit &= vmask;
        if (cx.ballot(hit != 0u) != 0ull)
            corr_emit_hits(cx, p, hit, vmask, x, xcorr, abits, kb, CF4_T);
        cx.sync();
    }
    // carry the last N stream samples to the next call (set_history(N+1), :95)
    if (seg == p.nseg - 1) {
        cf* ho = p.hist_out + (long)c * N;
        for (int j = t; j < N; j += CF4_T) {
            const int s = n - N + j;
            ho[j] = (s >= 0) ? xin[s] : hist[N + s];
        }
    }
}


} // namespace aisx
