// k_agc.h -- analog.feedforward_agc_cc(nsamples, reference) as the reference chain
// uses it (python/ais_demod.py:35,56).  [GR] feedforward_agc_cc_impl::work:
//   out[i] = reference / max(floor, max_{j<nsamples} envelope(in[i+j])) * in[i]
// (floor = 1e-4, AGC_FLOOR_DEFAULT: GNU Radio 3.7/3.8 "float max_env = 1e-4; // avoid divide by
// zero, indirectly set max gain"; the 1e-12 of the line upstream has commented out stays
// reachable through aisx_agc_set_floor)
// with history nsamples (so the output is the input delayed by nsamples-1 and the
// window looks ahead), envelope(x) = max(|re|,|im|) + 0.4*min(|re|,|im|) in double.
// The sliding maximum is computed exactly (max is associative) with a doubling
// table in LDS: M_k[i] = max(e[i .. i+2^k)), window W = max(M_K[i], M_K[i+W-2^K]).
#pragma once
#include "aisx_common.h"

namespace aisx {

constexpr int AGC_T = 256;       // threads per workgroup (k_agc)
constexpr int AGC_TL = 2048;     // outputs per tile
constexpr int AGC_PER = AGC_TL / AGC_T;
constexpr int AGC_MAXW = 2048;   // largest supported window
constexpr int AGC_E = AGC_TL + AGC_MAXW; // envelope slots per buffer
constexpr int AGC_LDS_BYTES = 2 * AGC_E * 4;

// two consecutive items; rows are only 8-byte aligned
struct __attribute__((packed, aligned(8))) cf_pair_agc { cf a, b; };

struct AgcParams {
    const cf* in; long in_stride;   // [nchan][n] new items
    cf* out; long out_stride;       // [nchan][n]
    const cf* hist_in; cf* hist_out; // [nchan][W-1]
    int n, W;
    float reference;
    float floor_env; // the initial max_env of [GR] feedforward_agc_cc_impl::work
    int ntiles;
    // Fused front end (agc8 only; phases == nullptr: the plain block).  The block's input is then
    // square_and_fft_sync_cc's output WITHOUT that output ever being stored: item m of it is
    // raw[m] * e^{j phases[m]} (python/gmsk_sync.py:26-28,33; the phases come from fs_walk_body,
    // k_freqsync.h), raw = the pending partial vector followed by the new samples `in`.
    // `phases` holds the phase of every NCO_CK-th item (phi[NCO_CK j], fs_walk_body's checkpoints), `dvec` each
    // vector's increment: the phases in between are walked again here (nco_phase_step, the walk's statement).
    const float* phases; long phases_stride; // [nchan][n / NCO_CK]
    const float* dvec; long dvec_stride;     // [nchan][n / 1024]
    const float* sintab;                     // gr::fxpt's sine table (NCO_TAB_FLOATS floats, aisx_tables.h)
    const cf* pend_in; cf* pend_out;         // [nchan][1024] pending partial vector in / out
    int npend, n_raw;                        // valid pending items; new raw samples per channel
};

constexpr float AGC_FLOOR_DEFAULT = 1e-4f;

AISX_HD float agc_envelope(cf x)
{
    const float r_abs = fabsf(x.re), i_abs = fabsf(x.im);
    float e;
    if (r_abs > i_abs)
        e = (float)((double)r_abs + 0.4 * (double)i_abs);
    else
        e = (float)((double)i_abs + 0.4 * (double)r_abs);
    // std::max(max_env, e) never selects a NaN; 0 is below any positive floor
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
            max_env = (p.floor_env < max_env) ? max_env : p.floor_env;
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

// The same for windows that are a multiple of 8 (the stock 512 is): each thread owns 8
// consecutive items.  The window of item i = 8a + k is the tail of its own group (a running
// maximum the thread keeps in registers), the Q - 1 = W/8 - 1 whole groups after it (a
// sliding maximum over GROUP maxima: 1/8 of the elements, log2(Q) doubling passes) and the
// first k items of group a + Q (that group's prefix maxima, exchanged through LDS).  About
// three LDS operations per item instead of three per item and doubling pass.
// Workgroups of 1024 threads, 8192 outputs per tile: the 511-item halo a tile reads (and, in the
// fused front end, mixes) a second time is 6 % of it (25 % with 256 threads: 2.26 -> 2.04 ms per
// launch in the chain at 512 threads, the step 1.7 % shorter at 1024; -DAGC8_THREADS rebuilds)
#ifndef AGC8_THREADS
#define AGC8_THREADS 1024
#endif
constexpr int AGC8_T = AGC8_THREADS;
constexpr int AGC8_G = 8;
constexpr int AGC8_TL = AGC8_G * AGC8_T;               // outputs per tile: one group per thread
constexpr int AGC8_NG = AGC8_TL / AGC8_G;               // groups with outputs per tile (= AGC8_T)
constexpr int AGC8_MAXQ = AGC_MAXW / AGC8_G;           // halo groups at most
constexpr int AGC8_GROUPS = AGC8_NG + AGC8_MAXQ;       // group maxima per buffer
constexpr int AGC8_LDS_BYTES = (2 * AGC8_GROUPS + AGC8_NG * AGC8_G) * 4; // group maxima x 2, prefix maxima of the NG groups windows end in
constexpr int AGC8_LDS_BYTES_MIXED = AGC8_LDS_BYTES + NCO_TAB_FLOATS * 4;  // + the NCO's sine table (fused front end)
static_assert(AGC8_NG == AGC8_T, "one output group per thread");
static_assert(AGC8_MAXQ <= AGC8_T, "at most one halo group per thread");

AISX_HD bool agc8_applies(int W) { return W % AGC8_G == 0 && W >= 2 * AGC8_G && W <= AGC_MAXW; }

template <class Ctx>
AISX_DI void agc8_body(Ctx& cx, const AgcParams& p)
{
    const int t = cx.tid();
    const int c = cx.by();
    const int tile = cx.bx();
    float* GA = (float*)cx.lds();            // group maxima, ping
    float* GB = GA + AGC8_GROUPS;            // ... pong
    float* PF = GB + AGC8_GROUPS;            // prefix maxima of the groups a window can end in, g = Q .. Q + NG - 1:
                                             // PF[(g - Q) * 8 + k] = max(e[8g .. 8g+k])
    const int H = p.W - 1;
    const int Q = p.W / AGC8_G;
    const int n = p.n;
    const cf* xin = p.in + (long)c * p.in_stride;
    const cf* hist = p.hist_in + (long)c * H;
    cf* xout = p.out + (long)c * p.out_stride;

    const int base = tile * AGC8_TL;
    const int nout = (n - base) < AGC8_TL ? (n - base) : AGC8_TL;
    const int E = (nout > 0 ? nout : 0) + H;        // items of the combined stream this tile looks at
    const int ngroups = (E + AGC8_G - 1) / AGC8_G;  // <= AGC8_NG + Q
    // fused front end: new item m is raw[m] mixed with the walked NCO phase (AgcParams)
    const bool mixed = p.phases != nullptr;
    const float* phi = mixed ? p.phases + (long)c * p.phases_stride : nullptr;
    const cf* pend = mixed ? p.pend_in + (long)c * 1024 : nullptr;
    float* ST = PF + AGC8_NG * AGC8_G; // the NCO's sine table (fused front end only)
    auto mix = [&](cf raw, float ph) -> cf {
        float sn, cs;
        nco_sincos(ph, ST, &sn, &cs);       // [GR] frequency_modulator_fc: gr::fxpt::sincos of d_phase
        return cmul_exact(raw, mk(cs, sn)); // multiply_cc(stream, frequency_modulator_fc output)
    };
    auto raw_item = [&](int m) -> cf { // raw sample m >= 0 behind the pending partial vector
        return (m < p.npend) ? pend[m] : xin[m - p.npend];
    };
    const float* dv = mixed ? p.dvec + (long)c * p.dvec_stride : nullptr;
    auto phase_at = [&](int m) -> float { // NCO phase of item m: from the checkpoint at or before it
        float ph = phi[m / NCO_CK];
        const float d = dv[m >> 10]; // (a checkpoint and the items behind it lie in one vector)
        const bool small = fabsf(d) < 6.0f;
#pragma nounroll
        for (int k = m % NCO_CK; k > 0; k--)
            ph = nco_phase_step(ph, d, small);
        return ph;
    };
    auto item = [&](int m) -> cf { // item m >= 0 of the block's input (needs the table in LDS)
        return mixed ? mix(raw_item(m), phase_at(m)) : xin[m];
    };

    bool allsmall = true; // every phase increment this wave's groups meet is small (|d| < 6: nco_wrap_small applies)
    // One group in two steps, so that in the fused build every global load of the tile is in
    // flight before the barrier that publishes the sine table: (1) load -- raw samples and, where
    // they are still to be mixed, their NCO phases (mask `mx`); (2) mix, envelopes, prefix maxima
    // to LDS, group maximum.
    struct Grp {
        cf v[AGC8_G];
        float f[AGC8_G];
        unsigned mx;
        float ck[AGC8_G / NCO_CK], d; // walk: f[] is still to be walked from the checkpoints with increment d (finish_group)
        bool walk;
    };
    auto load_group = [&](int g, Grp& G) {
        const int s0 = base + g * AGC8_G;
        const int m0 = s0 - H; // first item of the group in the block's input
        G.mx = 0;
        G.walk = false;
        G.d = 0.f;
#pragma unroll
        for (int i = 0; i < AGC8_G / NCO_CK; i++)
            G.ck[i] = 0.f;
        // wholly inside the new samples: 16-byte loads.  (A tile's last group looks at fewer than 8 items; where the
        // row goes on behind them it still takes this path -- the envelopes of the others are zeroed in finish_group.)
        const int avail = mixed ? p.npend + p.n_raw : n;
        if (s0 >= H && (g * AGC8_G + AGC8_G <= E || m0 + AGC8_G <= avail) && (!mixed || m0 >= p.npend)) {
            const cf_pair_agc* src = (const cf_pair_agc*)(xin + (m0 - (mixed ? p.npend : 0)));
#pragma unroll
            for (int k = 0; k < AGC8_G / 2; k++) {
                const cf_pair_agc q = src[k];
                G.v[2 * k] = q.a;
                G.v[2 * k + 1] = q.b;
            }
            if (mixed) {
                // (the window is a multiple of 8, so m0 = 1 mod 8: the checkpoint phi[m0 - 1] opens the group, every
                // NCO_CK-th item behind it is one, and phi[m0 + 7] is its last item -- possibly the first of the next vector)
                const int j = (m0 - 1) / NCO_CK;
                const float d = dv[(m0 - 1) >> 10];
#pragma unroll
                for (int i = 0; i < AGC8_G / NCO_CK; i++)
                    G.ck[i] = phi[j + i];
                G.f[7] = phi[j + AGC8_G / NCO_CK];
                G.d = d;
                G.walk = true;
                G.mx = 0xffu;
            }
        } else {
#pragma unroll
            for (int k = 0; k < AGC8_G; k++) {
                const int j = g * AGC8_G + k, s = base + j;
                G.v[k] = mk(0.f, 0.f);
                G.f[k] = 0.f;
                if (j < E) {
                    if (s < H)
                        G.v[k] = hist[s];
                    else if (!mixed)
                        G.v[k] = xin[s - H];
                    else {
                        G.v[k] = raw_item(s - H);
                        // (one step from the item before it where that one was walked too: a group at the edge of a
                        // tile must not cost its wave a walk per item)
                        const int m = s - H;
                        if (k > 0 && ((G.mx >> (k - 1)) & 1u) && m % NCO_CK != 0) {
                            const float d = dv[m >> 10];
                            G.f[k] = nco_phase_step(G.f[k - 1], d, fabsf(d) < 6.0f);
                        } else {
                            G.f[k] = phase_at(m);
                        }
                        G.mx |= 1u << k;
                    }
                }
            }
        }
    };
    cf own[AGC8_G];
    float sfx[AGC8_G]; // suffix maxima of the thread's output group: max(e[k .. 7])
    auto finish_group = [&](int g, bool keep, Grp& G) {
        cf (&v)[AGC8_G] = G.v;
        if (mixed) {
            if (G.walk) { // the phases between the checkpoints: the walk's statement again (k_freqsync.h); chains of NCO_CK - 1
                const float d = G.d;
                // item m0 + k is (k + 1) items behind the group's first checkpoint
                auto chains = [&](auto wrap) {
#pragma unroll
                    for (int k = 0; k < 7; k++) {
                        if ((k + 1) % NCO_CK == 0)
                            G.f[k] = G.ck[(k + 1) / NCO_CK];
                        else
                            G.f[k] = wrap((((k + 1) % NCO_CK == 1) ? G.ck[(k + 1) / NCO_CK] : G.f[k - 1]) + d);
                    }
                };
                if (allsmall) // (always, for estimates inside the band: the wrap as a select, no branch)
                    chains([](float x) { return nco_wrap_small(x); });
                else
                    chains([](float x) { return nco_wrap(x); });
            }
#pragma unroll
            for (int k = 0; k < AGC8_G; k++)
                if ((G.mx >> k) & 1u)
                    v[k] = mix(v[k], G.f[k]);
        }
        float e[AGC8_G];
#pragma unroll
        for (int k = 0; k < AGC8_G; k++)
            e[k] = (g * AGC8_G + k < E) ? agc_envelope(v[k]) : 0.f; // (0 never wins: the floor is positive)
        float run = e[0];
        float pfx[AGC8_G];
        pfx[0] = run;
#pragma unroll
        for (int k = 1; k < AGC8_G; k++) {
            run = run < e[k] ? e[k] : run;
            pfx[k] = run;
        }
        const int gp = g - Q;
        if (gp >= 0 && gp < AGC8_NG) {
#pragma unroll
            for (int k = 0; k < AGC8_G; k++)
                PF[gp * AGC8_G + k] = pfx[k];
        }
        GA[g] = run;
        if (keep) {
            float r2 = e[AGC8_G - 1];
            sfx[AGC8_G - 1] = r2;
#pragma unroll
            for (int k = AGC8_G - 2; k >= 0; k--) {
                r2 = r2 < e[k] ? e[k] : r2;
                sfx[k] = r2;
            }
#pragma unroll
            for (int k = 0; k < AGC8_G; k++)
                own[k] = v[k];
        }
    };
    {
        Grp G1, G2;
        const bool have1 = t < ngroups, have2 = AGC8_NG + t < ngroups;
        if (have1)
            load_group(t, G1);
        if (have2)
            load_group(AGC8_NG + t, G2);
        if (mixed) {
            typedef float f4 __attribute__((vector_size(16)));
            for (int i = t; i < NCO_TAB_FLOATS / 4; i += AGC8_T) // (8 KB, L2-resident: two 16-byte loads per thread)
                ((f4*)ST)[i] = ((const f4*)p.sintab)[i];
            cx.sync();
            // (decided here, behind the barrier the loads fly over and where all lanes are present; wave-uniform)
            const bool mine_small = (!have1 || !G1.walk || fabsf(G1.d) < 6.0f) && (!have2 || !G2.walk || fabsf(G2.d) < 6.0f);
            allsmall = cx.ballot(!mine_small) == 0ull;
        }
        if (have1)
            finish_group(t, true, G1);
        if (have2)
            finish_group(AGC8_NG + t, false, G2);
    }
    for (int g = ngroups + t; g < AGC8_GROUPS; g += AGC8_T)
        GA[g] = 0.f; // groups past the data: neutral
    cx.sync();
    // sliding maximum over Q - 1 group maxima by doubling: 2^K <= Q - 1
    int K = 0;
    while ((2 << K) <= Q - 1)
        K++;
    float* src = GA;
    float* dst = GB;
    for (int k = 0; k < K; k++) {
        const int step = 1 << k;
        for (int g = t; g < AGC8_GROUPS; g += AGC8_T) {
            const float a = src[g];
            const float b = (g + step < AGC8_GROUPS) ? src[g + step] : a;
            dst[g] = a < b ? b : a;
        }
        cx.sync();
        float* tmp = src;
        src = dst;
        dst = tmp;
    }
    if (nout > 0 && t * AGC8_G < nout) {
        // whole groups t+1 .. t+Q-1
        const float w1 = src[t + 1], w2 = src[t + Q - (1 << K)];
        const float gw = w1 < w2 ? w2 : w1;
        const float* pf = PF + t * AGC8_G; // prefix maxima of group t + Q, the one the window ends in
        cf o[AGC8_G];
#pragma unroll
        for (int k = 0; k < AGC8_G; k++) {
            float mx = sfx[k] < gw ? gw : sfx[k];
            if (k > 0) {
                const float pe = pf[k - 1];
                mx = mx < pe ? pe : mx;
            }
            mx = (p.floor_env < mx) ? mx : p.floor_env;
            const float gain = fdiv_rn(p.reference, mx);
            o[k] = mk(gain * own[k].re, gain * own[k].im);
        }
        const int i0 = t * AGC8_G;
        if (i0 + AGC8_G <= nout) {
            cf_pair_agc* dstp = (cf_pair_agc*)(xout + base + i0);
#pragma unroll
            for (int k = 0; k < AGC8_G / 2; k++) {
                cf_pair_agc q;
                q.a = o[2 * k];
                q.b = o[2 * k + 1];
                dstp[k] = q;
            }
        } else {
#pragma unroll
            for (int k = 0; k < AGC8_G; k++)
                if (i0 + k < nout)
                    xout[base + i0 + k] = o[k];
        }
    }
    // set_history(nsamples): keep the last W-1 items of the combined stream
    if (tile == p.ntiles - 1) {
        cf* ho = p.hist_out + (long)c * H;
        for (int j = t; j < H; j += AGC8_T) {
            const int s = n + j;
            ho[j] = (s < H) ? hist[s] : item(s - H);
        }
        if (mixed) { // stream_to_vector's pending items: the raw samples behind the last whole vector
            cf* po = p.pend_out + (long)c * 1024;
            const int rem = p.npend + p.n_raw - n;
            for (int i = t; i < rem; i += AGC8_T) {
                const int m = n + i;
                po[i] = (m < p.npend) ? pend[m] : xin[m - p.npend];
            }
        }
    }
}

} // namespace aisx
