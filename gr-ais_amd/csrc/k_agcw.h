// k_agcw.h -- analog.feedforward_agc_cc(512, reference) (python/ais_demod.py:35,56), optionally with
// square_and_fft_sync_cc's NCO mix in front of it (python/gmsk_sync.py:26-28,33), as a STREAMING
// kernel: no workgroup barrier in the data path, no LDS but the NCO's sine table.
//
// Why a second AGC kernel (round 5).  agc8_body (k_agc.h) is a bulk-synchronous tile: sixteen waves
// load, meet at a barrier, mix, meet, run six doubling passes over LDS with a barrier each, store.
// One such workgroup fits a CU, so loads, arithmetic and stores of a CU never overlap: 2.0 ms for
// 4.6 GB and ~0.5 ms of VALU work, and neither fewer instructions (round 3: -10 per sample) nor
// fewer bytes (round 4: -1.9 GB) moved it.  Here every wave is on its own: it walks consecutive
// 512-item blocks of one channel, the next block's samples in flight while this one is computed,
// and the sliding maximum needs nothing from any other wave.
//
// The sliding maximum over W = 512 items (van Herk / Gil-Werman, exact since max is associative):
// cut the stream into blocks of W.  The window of item o of block b is the tail [o, W) of its own
// block and the head [0, o) of block b + 1:
//     max_env(b, o) = max(S_b[o], P_{b+1}[o - 1]),  S = suffix maxima, P = prefix maxima of a block.
// A block is one wave: lane l owns the 8 consecutive items 8l .. 8l+7.  Prefix / suffix maxima
// inside a lane are register chains; across lanes two wave scans of the 64 group maxima (DPP);
// block b's items wait in registers until block b + 1's prefix maxima exist.  Envelopes are
// non-negative and never NaN (agcw_envelope), so every maximum is taken on the bit patterns as
// integers (same order, one v_max_i32 / v_max3_i32 each, no NaN canonicalisation).
//
// Block grid.  With H = W - 1 = 511 items of history the block's combined input is
//     s[j] = hist[j] (j < H),  s[H + m] = item m of the new input,   out[i] = gain_i * s[i], i < n.
// Blocks are aligned on m (the NCO's checkpoints phi[m / 8] then open every lane's group, and with
// nothing pending the loads are 64-byte aligned): block B holds m = 512 B + 8 l + k, i.e.
// j = m + 511; block -1 is the history (its first slot, j = -1, does not exist).  n is a multiple
// of 512: the new items are blocks 0 .. NB-1 exactly, outputs come from blocks -1 .. NB-2 plus
// the one item i = n - 1 that opens block NB-1 (its window is that whole block).  A wave takes
// AGW_RUN consecutive output blocks and reads one block more (1 / AGW_RUN of halo).
#pragma once
#include <type_traits>
#include "k_agc.h"

namespace aisx {

constexpr int AGW_W = 512;          // the window this kernel serves: 64 lanes x 8 items
constexpr int AGW_G = 8;            // items per lane
#ifndef AGW_WAVES_PER_WG
#define AGW_WAVES_PER_WG 4
#endif
constexpr int AGW_WAVES = AGW_WAVES_PER_WG; // waves per workgroup (they share nothing but the sine table)
constexpr int AGW_T = 64 * AGW_WAVES;
#ifndef AGW_RUN
#define AGW_RUN 16                  // output blocks per wave
#endif
constexpr int AGW_LDS_BYTES = NCO_TAB_FLOATS * 4;

// n new items (a multiple of the block) through a window of 512
AISX_HD bool agcw_applies(int W, int n) { return W == AGW_W && n >= AGW_W && n % AGW_W == 0; }
AISX_HD int agcw_grid(int n) { return ((n / AGW_W + AGW_RUN - 1) / AGW_RUN + AGW_WAVES - 1) / AGW_WAVES; }

AISX_HD int f2i(float x)
{
    int r;
    __builtin_memcpy(&r, &x, 4);
    return r;
}
AISX_HD float i2f(int x)
{
    float r;
    __builtin_memcpy(&r, &x, 4);
    return r;
}
AISX_HD int imax(int a, int b) { return a < b ? b : a; }

// gr::fxpt::float_to_fixed of an NCO phase.  frequency_modulator_fc's wrap leaves phases in
// (-3 pi, pi): a negative increment walks down from -pi to -3 pi before fmod brings it back, so the
// fold of fxpt_float_to_fixed (aisx_common.h) is live for half of all channels.  For -9.4 <= x < -pi
// its d = floor(x / 2 pi + 0.5) is -1 (x / 2 pi in [-1.4961, -0.5), and the quotient of any float
// below -pi rounds below -0.5) and x -= d * 2 pi is the one float addition x + 2 pi; for
// -pi <= x < pi it does nothing.  nco_fold_fast is that; nco_fold_ok says whether x lies in
// [-9.4, pi) (one v_med3 + one compare: the upper bound is the float below pi); anything else takes
// the general statement (nco_fold_general).
AISX_HD float nco_fold_fast(float x)
{
    const float PI = 3.14159265358979323846f, TAU = 2.0f * 3.14159265358979323846f;
    const float y = x + TAU;
    return (x < -PI) ? y : x;
}
AISX_HD bool nco_fold_ok(float x)
{
    const float BELOW_PI = 3.14159250259399414062f; // nextafterf(pi, 0)
    const float lo = -9.4f;
    // median of (x, lo, BELOW_PI) == x  <=>  lo <= x <= BELOW_PI  (a NaN compares unequal)
    const float m = fmaxf(fminf(x, BELOW_PI), lo);
    return m == x;
}
AISX_HD float nco_fold_general(float x)
{
    const float PI = 3.14159265358979323846f, TAU = 2.0f * 3.14159265358979323846f;
    if (!(x >= -PI && x < PI)) {
        const int d = (int)floor((double)fdiv_rn(x, TAU) + 0.5);
        x -= (float)d * TAU;
    }
    return x;
}
// (int)(x * 2^31 / pi) of a folded phase (division by pi: aisx_common.h fxpt_float_to_fixed)
AISX_HD int nco_folded_to_fixed(float x)
{
    const float PI = 3.14159265358979323846f, TWO_TO_THE_31 = 2147483648.0f;
    const float RCP_PI = 0.318309886183790671538f;
    const float y = x * TWO_TO_THE_31;
    const float q = y * RCP_PI;
    return (int)fmaf(fmaf(-q, PI, y), RCP_PI, q);
}
AISX_HD int nco_phase_to_fixed(float x) { return nco_folded_to_fixed(nco_fold_ok(x) ? nco_fold_fast(x) : nco_fold_general(x)); }

// the gain reference / max_env.  With a power-of-two reference (the stock 2.0) it is the correctly
// rounded reciprocal scaled exactly: v_rcp_f32 (1 ulp) and one Newton step give that reciprocal for
// every max_env in [2^-100, 2^100] -- settled exhaustively on the device (round 3; the sweep is
// aisx_util_agc_rcp_mismatches, tests/test_gpu_stages.py runs it again) -- four instructions where
// the division sequence takes eleven.  Anything else divides.
constexpr float AGW_RCP_LO = 7.888609052210118e-31f, AGW_RCP_HI = 1.2676506002282294e30f; // 2^-100, 2^100
template <class Ctx>
AISX_DI float agcw_gain_fast(const Ctx& cx, float reference, float max_env)
{
    const float r0 = cx.rcp_approx(max_env);
    const float e = fmaf(-max_env, r0, 1.0f);
    const float r1 = fmaf(e, r0, r0);
    return r1 * reference; // (an exact scaling)
}
// is `reference` a power of two whose scaling of a reciprocal in [2^-100, 2^100] can neither overflow nor go subnormal?
AISX_HD bool agcw_fast_reference(float reference)
{
    int ex = 0;
    const float m = frexpf(reference, &ex);
    return m == 0.5f && ex > -20 && ex < 20;
}

// envelope(x) of feedforward_agc_cc with a NaN mapped to 0 (std::max never selects it; 0 is below any
// positive floor): the IEEE maximum of (e, 0) is that in one instruction (e >= 0 otherwise)
AISX_HD float agcw_envelope(cf x)
{
    const float r_abs = fabsf(x.re), i_abs = fabsf(x.im);
    const bool rg = r_abs > i_abs;
    const float big = rg ? r_abs : i_abs, small = rg ? i_abs : r_abs;
    const float e = (float)((double)big + 0.4 * (double)small);
    return fmaxf(e, 0.0f);
}

template <bool MIXED, class Ctx>
AISX_DI void agcw_body(Ctx& cx, const AgcParams& p)
{
    const int t = cx.tid();
    const int l = t & 63;
    const int wave = cx.wave_id();
    const int c = cx.by();
    const float* ST = (const float*)cx.lds();
    if (MIXED) {
        typedef float f4 __attribute__((vector_size(16)));
        float* st = (float*)cx.lds();
        for (int i = t; i < NCO_TAB_FLOATS / 4; i += AGW_T) // (8 KB, L2-resident)
            ((f4*)st)[i] = ((const f4*)p.sintab)[i];
        cx.sync(); // the only workgroup barrier: every wave is on its own from here
    }
    const int n = p.n;
    const int NB = n / AGW_W;
    const int ob0 = (cx.bx() * AGW_WAVES + wave) * AGW_RUN; // output blocks ob0 .. ob1-1, block B = ob - 1
    if (ob0 >= NB)
        return;
    const int ob1 = (ob0 + AGW_RUN < NB) ? ob0 + AGW_RUN : NB;
    const bool last_run = ob1 == NB;
    const int B0 = ob0 - 1, Bend = ob1 - 1; // blocks B0 .. Bend are read, B0 .. Bend - 1 give outputs

    constexpr int H = AGW_W - 1;
    const cf* xin = p.in + (long)c * p.in_stride;
    const cf* hist = p.hist_in + (long)c * H;
    cf* xout = p.out + (long)c * p.out_stride;
    const int npend = MIXED ? p.npend : 0;
    const cf* pend = MIXED ? p.pend_in + (long)c * 1024 : nullptr;
    const float* phi = MIXED ? p.phases + (long)c * p.phases_stride : nullptr;
    const float* dv = MIXED ? p.dvec + (long)c * p.dvec_stride : nullptr;
    const int floor_i = f2i(p.floor_env);
    // (wave-uniform) the reciprocal form of the gain serves this call: reference and floor in its range
    const bool fast_ref = agcw_fast_reference(p.reference) && p.floor_env >= AGW_RCP_LO && p.floor_env <= AGW_RCP_HI;
    const int rcp_hi_i = f2i(AGW_RCP_HI);

    // what a block's lane loads: its 8 raw items (history: already mixed), the NCO checkpoint that
    // opens the group and the increment of the group's vector (a block lies in one 1024-vector)
    struct Raw {
        cf v[AGW_G];
        float ck, d;
    };
    // INTERIOR: the block lies wholly in the new samples (16-byte loads); otherwise any block
    auto load = [&](auto INTERIOR, int B, Raw& R) {
        constexpr bool interior = decltype(INTERIOR)::value;
        R.ck = 0.f;
        R.d = 0.f;
        if (!interior && B < 0) { // the history: j = 8 l + k - 1
#pragma unroll
            for (int k = 0; k < AGW_G; k++) {
                const int j = AGW_G * l + k - 1;
                R.v[k] = (j >= 0) ? hist[j] : mk(0.f, 0.f);
            }
            return;
        }
        const int m0 = AGW_W * B + AGW_G * l;
        if (interior || AGW_W * B >= npend) { // (wave-uniform)
            const cf_pair_agc* src = (const cf_pair_agc*)(xin + (m0 - npend));
#pragma unroll
            for (int k = 0; k < AGW_G / 2; k++) {
                const cf_pair_agc q = src[k];
                R.v[2 * k] = q.a;
                R.v[2 * k + 1] = q.b;
            }
        } else { // the pending partial vector comes first
#pragma unroll
            for (int k = 0; k < AGW_G; k++) {
                const int m = m0 + k;
                R.v[k] = (m < npend) ? pend[m] : xin[m - npend];
            }
        }
        if (MIXED) {
            R.ck = phi[m0 / NCO_CK];
            R.d = dv[B >> 1];
        }
    };

    // a block with its items mixed, their envelopes' prefix / suffix maxima inside the lane (bit
    // patterns), the maxima over all lanes before / behind this one and over the whole wave
    struct Blk {
        cf v[AGW_G];
        int pfx[AGW_G], sfx[AGW_G];
        int xp, xs, all;
    };
    auto finish = [&](auto INTERIOR, int B, const Raw& R, Blk& K) {
        constexpr bool interior = decltype(INTERIOR)::value;
        float e[AGW_G];
        if (MIXED && (interior || B >= 0)) {
            // phases of the group: the checkpoint is item 0's (m0 is a multiple of NCO_CK = 8), the others are
            // walked again with the walk's own statement (fs_walk_body)
            static_assert(NCO_CK == AGW_G, "one checkpoint opens each lane's group");
            float f[AGW_G];
            f[0] = R.ck;
            const float d = R.d;
            if (fabsf(d) < 6.0f) { // (wave-uniform: one vector, one increment)
#pragma unroll
                for (int k = 1; k < AGW_G; k++)
                    f[k] = nco_wrap_small(f[k - 1] + d);
            } else {
#pragma unroll
                for (int k = 1; k < AGW_G; k++)
                    f[k] = nco_wrap(f[k - 1] + d);
            }
            // float_to_fixed's fold: the one-addition form for the whole wave unless some lane holds a phase
            // outside [-9.4, pi) (one decision per block, not eight divergent ones)
            bool ok = true;
#pragma unroll
            for (int k = 0; k < AGW_G; k++)
                ok = ok && nco_fold_ok(f[k]);
            if (cx.ballot(!ok) == 0ull) {
#pragma unroll
                for (int k = 0; k < AGW_G; k++)
                    f[k] = nco_fold_fast(f[k]);
            } else {
#pragma unroll
                for (int k = 0; k < AGW_G; k++)
                    f[k] = nco_fold_general(f[k]);
            }
#pragma unroll
            for (int k = 0; k < AGW_G; k++) {
                // [GR] frequency_modulator_fc: gr::fxpt::sincos(float_to_fixed(d_phase)); multiply_cc
                const unsigned x = (unsigned)nco_folded_to_fixed(f[k]);
                const unsigned xc = x + 0x40000000u;
                const cf es = cx.lds_cf(ST, x >> 22), ec = cx.lds_cf(ST, xc >> 22);
                const float sn = es.re * (float)(x >> 1) + es.im;
                const float cs = ec.re * (float)(xc >> 1) + ec.im;
                K.v[k] = cmul_exact(R.v[k], mk(cs, sn));
            }
        } else {
#pragma unroll
            for (int k = 0; k < AGW_G; k++)
                K.v[k] = R.v[k];
        }
#pragma unroll
        for (int k = 0; k < AGW_G; k++)
            e[k] = agcw_envelope(K.v[k]);
        if (!interior && B < 0 && l == 0)
            e[0] = 0.f; // j = -1: before the stream (0 never wins: the floor is positive)
        K.pfx[0] = f2i(e[0]);
#pragma unroll
        for (int k = 1; k < AGW_G; k++)
            K.pfx[k] = imax(K.pfx[k - 1], f2i(e[k]));
        K.sfx[AGW_G - 1] = f2i(e[AGW_G - 1]);
#pragma unroll
        for (int k = AGW_G - 2; k >= 0; k--)
            K.sfx[k] = imax(K.sfx[k + 1], f2i(e[k]));
        K.xp = cx.wave_excl_prefix_max_nn(K.pfx[AGW_G - 1]);
        K.xs = cx.wave_excl_suffix_max_nn(K.pfx[AGW_G - 1], K.all);
    };
    // outputs of block B (all of its items) given the block behind it
    auto emit = [&](auto INTERIOR, int B, const Blk& K, const Blk& Nx) {
        constexpr bool interior = decltype(INTERIOR)::value;
        const int a = imax(K.xs, Nx.xp); // the 63 whole groups between the item's own and the window's last
        int mx[AGW_G];
#pragma unroll
        for (int k = 0; k < AGW_G; k++) {
            mx[k] = imax(K.sfx[k], a);
            if (k > 0)
                mx[k] = imax(mx[k], Nx.pfx[k - 1]);
            mx[k] = imax(mx[k], floor_i);
        }
        cf o[AGW_G];
        if (fast_ref && imax(K.all, Nx.all) <= rcp_hi_i) { // (wave-uniform)
#pragma unroll
            for (int k = 0; k < AGW_G; k++) {
                const float g = agcw_gain_fast(cx, p.reference, i2f(mx[k]));
                o[k] = mk(g * K.v[k].re, g * K.v[k].im);
            }
        } else {
#pragma unroll
            for (int k = 0; k < AGW_G; k++) {
                const float g = fdiv_rn(p.reference, i2f(mx[k]));
                o[k] = mk(g * K.v[k].re, g * K.v[k].im);
            }
        }
        const int i0 = AGW_W * B + H + AGW_G * l;
        if (interior || B >= 0) {
            cf_pair_agc* dst = (cf_pair_agc*)(xout + i0);
#pragma unroll
            for (int k = 0; k < AGW_G / 2; k++) {
                cf_pair_agc q;
                q.a = o[2 * k];
                q.b = o[2 * k + 1];
                dst[k] = q;
            }
        } else {
#pragma unroll
            for (int k = 0; k < AGW_G; k++)
                if (i0 + k >= 0)
                    xout[i0 + k] = o[k];
        }
    };
    // One step at block X: load block X + 1, finish block X, emit block X - 1.  `mine` raw / block belong to X,
    // `other` to X - 1 before the step and to X + 1 (raw) after it.
    auto step = [&](auto INTERIOR, int X, Raw& r_mine, Raw& r_other, Blk& k_mine, const Blk& k_other) {
        if (X + 1 <= Bend)
            load(INTERIOR, X + 1, r_other);
        finish(INTERIOR, X, r_mine, k_mine);
        if (X > B0)
            emit(INTERIOR, X - 1, k_other, k_mine);
    };
    std::integral_constant<bool, false> ANY;
    std::integral_constant<bool, true> INNER;

    Raw rA, rB;
    Blk kA, kB;
    int X = B0;
    load(ANY, X, rA);
    // steps that touch the history, the pending items or the run's first block: one copy of the code, blocks
    // handed on by assignment (a wave takes one, the first run of a channel two to four)
    while (X <= Bend && !(X > B0 && X >= 1 && AGW_W * (X + 1) >= npend)) {
        step(ANY, X, rA, rB, kA, kB);
        rA = rB;
        kB = kA;
        X++;
    }
    // (kB = block X - 1, rA = raw of block X)  interior steps in pairs: the two sets of registers swap roles
    bool final_in_A = false; // where the block finished last stands: kB after the loop above
    while (X <= Bend) {
        step(INNER, X, rA, rB, kA, kB);
        X++;
        final_in_A = true;
        if (X > Bend)
            break;
        step(INNER, X, rB, rA, kB, kA);
        X++;
        final_in_A = false;
    }
    // the block finished last is NB - 1: its first item is output n - 1 (window = the whole block), the
    // others are the history the next call starts from (set_history(nsamples))
    auto epilogue = [&](const Blk& cur) {
        if (l == 0) {
            const int mx = imax(imax(cur.sfx[0], cur.xs), floor_i);
            const float g = fdiv_rn(p.reference, i2f(mx));
            xout[n - 1] = mk(g * cur.v[0].re, g * cur.v[0].im);
        }
        cf* ho = p.hist_out + (long)c * H;
#pragma unroll
        for (int k = 0; k < AGW_G; k++) {
            const int j = AGW_G * l + k - 1;
            if (j >= 0)
                ho[j] = cur.v[k];
        }
        if (MIXED) { // stream_to_vector's pending items: the raw samples behind the last whole vector
            cf* po = p.pend_out + (long)c * 1024;
            const int rem = p.npend + p.n_raw - n;
            for (int i = l; i < rem; i += 64) {
                const int m = n + i;
                po[i] = (m < npend) ? pend[m] : xin[m - npend];
            }
        }
    };
    if (last_run) { // (two calls, not a choice of block: the blocks are registers)
        if (final_in_A)
            epilogue(kA);
        else
            epilogue(kB);
    }
}

} // namespace aisx
