// k_pfb.h -- wideband front end (BASELINE config 5): an M = 1024
// lane polyphase channelizer.  The reference builds one
// freq_xlating_fir_filter_ccf(decim, low_pass(1, rate, 11e3, 1e3), f_off, rate) per
// channel on the CPU (python/radio.py:49-54); for M uniformly spaced centre
// frequencies f_m = m*fs/M that bank is, exactly,
//   y_m[k] = e^{-j 2 pi m k D / M} * sum_p e^{+j 2 pi m p / M} u_k[p],
//   u_k[p] = sum_q h[p + q M] x[k D - p - q M]          (p = 0..M-1)
// i.e. a K-tap polyphase FIR per branch followed by an M-point inverse DFT across
// the branches.  One wave produces one output frame: lane l owns branches
// p = l + 64*n1, which is exactly the input layout of the 16 x 16 x 4 register /
// LDS FFT, so the partial sums never leave the VGPRs before the transform.
#pragma once
#include "aisx_common.h"
#include "k_fft.h"

namespace aisx {

constexpr int PFB_M = 1024;
constexpr int PFB_T = 256; // 4 waves = 4 frames per workgroup
constexpr int PFB_ROW = 68;
constexpr int PFB_WAVE_ELEMS = 16 * PFB_ROW;
constexpr int PFB_LDS_BYTES = (4 * PFB_WAVE_ELEMS + 64) * 8;

struct PfbParams {
    const cf* in; long in_stride;     // [nstreams][n] new wideband samples
    const cf* hist_in; cf* hist_out;  // [nstreams][Lh] previous samples (Lh = K*M)
    const float* taps;                // [K*M] prototype low-pass, zero padded
    const cf* wtab;                   // [1024] W_1024^k
    cf* out; long out_stride;         // [nstreams*M][out_stride]: lane m of stream s is row s*M + m
    int n, D, K, Lh, nframes;
    long frame0;                      // index of the first frame of this call (for the (-1)^(m k) type rotation)
};

template <class Ctx>
AISX_DI void pfb_body(Ctx& cx, const PfbParams& p)
{
    const int t = cx.tid();
    const int wave = t >> 6, l = t & 63;
    const int s = cx.by();
    const int f = cx.bx() * 4 + wave; // frame of this call
    cf* lds = (cf*)cx.lds();
    cf* X = lds + wave * PFB_WAVE_ELEMS;
    cf* T2 = lds + 4 * PFB_WAVE_ELEMS;
    if (t < 64)
        T2[t] = p.wtab[(16 * (t >> 2) * (t & 3)) & (PFB_M - 1)];
    const bool live = f < p.nframes;
    const cf* xin = p.in + (long)s * p.in_stride;
    const cf* hist = p.hist_in + (long)s * p.Lh;
    // newest sample of frame f is (call-relative) index e = f*D, as in GNU Radio's decimating
    // FIR (output k uses x[kD], x[kD-1], ...); branch p, tap q reads e - p - q*M
    const long e = (long)f * p.D;
    cf x[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; n1++) {
        const int br = l + 64 * n1;
        float ar = 0.f, ai = 0.f;
        if (live) {
            for (int q = 0; q < p.K; q++) {
                const long idx = e - br - (long)q * PFB_M;
                const cf v = (idx >= 0) ? xin[idx] : hist[p.Lh + idx];
                const float h = p.taps[br + q * PFB_M];
                ar = fmaf(h, v.re, ar);
                ai = fmaf(h, v.im, ai);
            }
        }
        x[n1] = mk(ar, ai);
    }
    // inverse 1024-point DFT across the branches (e^{+j}): DIF, 16 x 16 x 4
    dft16<true>(cx, x);
#pragma unroll
    for (int k1 = 1; k1 < 16; k1++)
        x[k1] = cmul_conj_fma(x[k1], p.wtab[(k1 * l) & (PFB_M - 1)]);
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++)
        X[k1 * PFB_ROW + l] = x[k1];
    cx.sync();
    {
        const int k1 = l >> 2, n3 = l & 3;
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++)
            x[n2] = X[k1 * PFB_ROW + n2 * 4 + n3];
        dft16<true>(cx, x);
#pragma unroll
        for (int k2 = 1; k2 < 16; k2++)
            x[k2] = cmul_conj_fma(x[k2], T2[k2 * 4 + n3]);
#pragma unroll
        for (int k2 = 0; k2 < 16; k2++)
            X[k1 * PFB_ROW + k2 * 4 + n3] = x[k2];
    }
    cx.sync();
    const long kabs = p.frame0 + f;
#pragma unroll
    for (int h = 0; h < 4; h++) {
        const int q = l + 64 * h, k1 = q >> 4, k2 = q & 15;
        cf y0 = X[k1 * PFB_ROW + k2 * 4 + 0], y1 = X[k1 * PFB_ROW + k2 * 4 + 1];
        cf y2 = X[k1 * PFB_ROW + k2 * 4 + 2], y3 = X[k1 * PFB_ROW + k2 * 4 + 3];
        dft4<true>(cx, y0, y1, y2, y3);
        const int mb = k1 + 16 * k2; // lane m = mb + 256*k3
        cf ys[4] = { y0, y1, y2, y3 };
#pragma unroll
        for (int k3 = 0; k3 < 4; k3++) {
            const int m = mb + 256 * k3;
            // e^{-j 2 pi m k D / M}: (m*k*D) mod M indexes the forward twiddle table
            const long r = ((long)m * (kabs % PFB_M) % PFB_M) * p.D % PFB_M;
            const cf rot = p.wtab[r];
            if (live)
                p.out[((long)s * PFB_M + m) * p.out_stride + f] = cmul_fma(ys[k3], rot);
        }
    }
    // history for the next call: last Lh samples of (hist ++ in), by the last workgroup row
    if (cx.bx() == 0) {
        cf* ho = p.hist_out + (long)s * p.Lh;
        for (int j = t; j < p.Lh; j += PFB_T) {
            const long idx = (long)p.n - p.Lh + j;
            ho[j] = (idx >= 0) ? xin[idx] : hist[p.Lh + idx];
        }
    }
}

} // namespace aisx
