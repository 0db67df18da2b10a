// k_agc.h -- analog.feedforward_agc_cc(nsamples, reference) as the reference chain
// uses it (python/ais_demod.py:35,56).  [GR] feedforward_agc_cc_impl::work:
//   out[i] = reference / max(1e-12, max_{j<nsamples} envelope(in[i+j])) * in[i]
// with history nsamples (so the output is the input delayed by nsamples-1 and the
// window looks ahead), envelope(x) = max(|re|,|im|) + 0.4*min(|re|,|im|) in double.
// The sliding maximum is computed exactly (max is associative) with a doubling
// table in LDS: M_k[i] = max(e[i .. i+2^k)), window W = max(M_K[i], M_K[i+W-2^K]).
#pragma once
#include "aisx_common.h"

namespace aisx {

constexpr int AGC_T = 256;       // threads per workgroup
constexpr int AGC_TL = 2048;     // outputs per tile
constexpr int AGC_PER = AGC_TL / AGC_T;
constexpr int AGC_MAXW = 2048;   // largest supported window
constexpr int AGC_E = AGC_TL + AGC_MAXW; // envelope slots per buffer
constexpr int AGC_LDS_BYTES = 2 * AGC_E * 4;

struct AgcParams {
    const cf* in; long in_stride;   // [nchan][n] new items
    cf* out; long out_stride;       // [nchan][n]
    const cf* hist_in; cf* hist_out; // [nchan][W-1]
    int n, W;
    float reference;
    int ntiles;
};

AISX_HD float agc_envelope(cf x)
{
    const float r_abs = fabsf(x.re), i_abs = fabsf(x.im);
    float e;
    if (r_abs > i_abs)
        e = (float)((double)r_abs + 0.4 * (double)i_abs);
    else
        e = (float)((double)i_abs + 0.4 * (double)r_abs);
    // std::max(max_env, e) never selects a NaN; 0 is below the 1e-12 floor
    return (e != e) ? 0.0f : e;
}

template <class Ctx>
AISX_DI void agc_body(Ctx& cx, const AgcParams& p)
{
    const int t = cx.tid();
    const int c = cx.by();
    const int tile = cx.bx();
    float* A = (float*)cx.lds();
    float* B = A + AGC_E;
    const int H = p.W - 1;
    const int n = p.n;
    const cf* xin = p.in + (long)c * p.in_stride;
    const cf* hist = p.hist_in + (long)c * H;
    cf* xout = p.out + (long)c * p.out_stride;

    const int base = tile * AGC_TL;        // first output index of the tile
    const int nout = (n - base) < AGC_TL ? (n - base) : AGC_TL;
    const int E = nout + H;                // envelopes needed: stream[base .. base+nout+H)
    // combined stream s[j] = hist[j] (j < H) else in[j - H]; out[i] uses s[i .. i+W)
    cf own[AGC_PER];
#pragma unroll
    for (int m = 0; m < AGC_PER; m++) {
        const int j = t + AGC_T * m;
        cf v = mk(0.f, 0.f);
        if (j < E) {
            const int s = base + j;
            v = (s < H) ? hist[s] : xin[s - H];
            A[j] = agc_envelope(v);
        }
        own[m] = v;
    }
    for (int j = AGC_TL + t; j < E; j += AGC_T) { // halo beyond the tile's own outputs
        const int s = base + j;
        const cf v = (s < H) ? hist[s] : xin[s - H];
        A[j] = agc_envelope(v);
    }
    cx.sync();
    int K = 0;
    while ((2 << K) <= p.W)
        K++; // 2^K <= W < 2^(K+1)
    float* src = A;
    float* dst = B;
    for (int k = 0; k < K; k++) {
        const int step = 1 << k;
        for (int j = t; j < E; j += AGC_T) {
            const float a = src[j];
            const float b = (j + step < E) ? src[j + step] : a;
            dst[j] = a < b ? b : a;
        }
        cx.sync();
        float* tmp = src;
        src = dst;
        dst = tmp;
    }
    const int shift = p.W - (1 << K);
#pragma unroll
    for (int m = 0; m < AGC_PER; m++) {
        const int i = t + AGC_T * m;
        if (i < nout) {
            const float a = src[i], b = src[i + shift];
            float max_env = a < b ? b : a;
            max_env = (1e-12f < max_env) ? max_env : 1e-12f;
            const float gain = fdiv_rn(p.reference, max_env);
            xout[base + i] = mk(gain * own[m].re, gain * own[m].im);
        }
    }
    // set_history(nsamples): keep the last W-1 items of the combined stream
    if (tile == p.ntiles - 1) {
        cf* ho = p.hist_out + (long)c * H;
        for (int j = t; j < H; j += AGC_T) {
            const int s = n + j; // combined index of the j-th kept item
            ho[j] = (s < H) ? hist[s] : xin[s - H];
        }
    }
}

} // namespace aisx
