// k_corr.h -- corr_est_cc on the GPU (reference: lib/corr_est_cc_impl.cc).
//
//   corr_inith_body   : forward transform of the (1/F-scaled, zero padded) taps,
//                       left in the FFT's own position order          [init]
//   corr_main_body    : A2 pass-through (delay N) + A3 overlap-save FFT
//                       correlation + A4 mag^2 + threshold bitmask     [hot]
//   corr_resolve_body : A5 peak search / centre of mass / phase / tags [sparse]
//
// FFT: F = 2048 = 16 x 16 x 8, 128 threads (2 waves) per transform, 16 points
// per thread in VGPRs, 2 LDS exchanges per direction.  Forward is DIF (natural
// in, digit-reversed positions out), the template spectrum H is stored in those
// same positions, inverse is DIT (natural out): no reordering pass.
// The template spectrum H (16 KB) and the twiddle table W_2048 (16 KB) stay in
// L2 and are read where they are used: keeping them in LDS / VGPRs costs
// occupancy (35 KB LDS, 225 VGPRs => 2 waves/SIMD; now 18 KB, 132 VGPRs => 3).
// LDS image: 16 rows x 136 complex (row pad 8 => the radix-16 passes over
// stride-8 columns hit 64 distinct banks per half wave) with the 16-byte chunk
// index XOR-ed by (k2>>2)&3 so the stride-1 radix-8 pass reads ds_read_b128
// conflict free.
#pragma once
#include <type_traits>
#include "aisx_common.h"
#include "k_fft.h"

namespace aisx {

constexpr int CF_F = 2048;    // FFT size
constexpr int CF_T = 128;     // threads per transform
constexpr int CF_ROW = 136;   // LDS row pitch in complex elements
constexpr int CF_LDS_ELEMS = 16 * CF_ROW + 128; // data + tw2 (H stays in L2: 16 KB, read per tile)
constexpr int CF_LDS_BYTES = CF_LDS_ELEMS * 8;

AISX_HD int cf_pos(int row, int col)
{
    int k2 = col >> 3, n3 = col & 7;
    return row * CF_ROW + (k2 << 3) + ((((n3 >> 1) ^ ((k2 >> 2) & 3)) << 1) | (n3 & 1));
}

struct CorrParams {
    const cf* in;   long in_stride;   // [nchan][n] new samples
    cf* out;        long out_stride;  // [nchan][n] delayed pass-through
    cf* corr;       long corr_stride; // correlator output (dense) or sparse scratch
    int dense_corr;
    const cf* hist_in;                // [nchan][N] last N samples of the previous call
    cf* hist_out;                     // [nchan][N]
    const cf* Hpos;                   // [F]
    const cf* wtab;                   // [F] W_F^k = exp(-2 pi j k / F)
    unsigned long long* abits; long abits_stride; // 1 bit per output item: !(mag <= thresh)
    int n, N, L, nseg, tiles_per_seg;
    float thresh;
    // first call after set_symbols(): the reference's FFT filter starts from a zeroed tail
    // ([GR] fft_filter_ccc::set_taps), i.e. the correlation sees zeros before this call's first
    // item, while the delayed pass-through still comes from the block's history (:180-184)
    int corr_hist_zero;
};

// The rare path of the main kernels' epilogue: this wave has at least one value above the
// threshold.  Hits go to the sparse scratch and to the hit bitmask.  The resolver's 3-point centre
// of mass (:219-227) also wants the two neighbours of a peak, which are usually below the
// threshold: consecutive lanes hold consecutive outputs (value n1 of thread t is element
// m = t + T n1 of the tile's transform, T a multiple of 64), so a lane whose neighbour lane has a
// hit leaves its own value in the scratch too.  Which neighbours that covers follows from the
// position alone -- element m +- 1 exists in the tile (N <= m +- 1 < F) and in the same wave
// ((m & 63) != 63 resp. != 0) -- and the resolver applies the same rule (corr_resolve_body); a
// neighbour in another wave or tile is recomputed there in direct form.
template <class Ctx>
AISX_DI void corr_emit_hits(Ctx& cx, const CorrParams& p, unsigned hit, unsigned vmask, const cf (&x)[16], cf* xcorr,
                            unsigned long long* abits, int kb, int stride)
{
    // (the hit masks of lane - 1 and lane + 1; 0 past the ends of the wave)
    const unsigned nb = (cx.lane_prev_u32(hit) | cx.lane_next_u32(hit)) & vmask & ~hit;
    // (One atomic per hit lane.  The hit bits of a slice as one ballot and two atomics per wave:
    // 1.31 against 1.28 ms per launch on the chain's data, DESIGN_APPENDIX.md A.)
#pragma unroll
    for (int n1 = 0; n1 < 16; n1++) {
        if (((hit | nb) >> n1) & 1u) {
            const int k = kb + stride * n1;
            if (!p.dense_corr)
                xcorr[k] = x[n1];
            if ((hit >> n1) & 1u)
                cx.atomic_or64(&abits[k >> 6], 1ull << (k & 63));
        }
    }
}

// ---- forward passes shared by init and main -------------------------------
// On entry x[n1] = w[t + 128*n1].  On exit v[h][k3] holds the spectrum at
// position (q = t + 128*h, k3), i.e. logical index q*8 + k3.
template <class Ctx>
AISX_DI void cf_forward(Ctx& cx, cf (&x)[16], const cf* wtab, cf* ldsX, const cf* ldsT, cf (&v)[2][8])
{
    const int t = cx.tid();
    dft16<false>(cx, x);
    // W_2048^{k1*t}: gathered from the 16 KB table (L1/L2 resident) where it is used --
    // holding the 15 values in VGPRs across the tile loop costs a wave of occupancy
#pragma unroll
    for (int k1 = 1; k1 < 16; k1++)
        x[k1] = cmul_fma(x[k1], wtab[k1 * t]);
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++)
        ldsX[cf_pos(k1, t)] = x[k1];
    cx.sync();
    {
        const int k1 = t >> 3, n3 = t & 7;
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++)
            x[n2] = ldsX[cf_pos(k1, n2 * 8 + n3)];
        dft16<false>(cx, x);
#pragma unroll
        for (int k2 = 1; k2 < 16; k2++)
            x[k2] = cmul_fma(x[k2], ldsT[k2 * 8 + n3]);
#pragma unroll
        for (int k2 = 0; k2 < 16; k2++)
            ldsX[cf_pos(k1, k2 * 8 + n3)] = x[k2];
    }
    cx.sync();
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int q = t + CF_T * h, k1 = q >> 4, k2 = q & 15, swz = (k2 >> 2) & 3;
        const int base = k1 * CF_ROW + k2 * 8;
#pragma unroll
        for (int pr = 0; pr < 4; pr++) {
            const int ch = base + 2 * (pr ^ swz);
            v[h][2 * pr] = ldsX[ch];
            v[h][2 * pr + 1] = ldsX[ch + 1];
        }
        dft8<false>(cx, v[h]);
    }
}

struct CorrInitParams {
    const cf* taps_scaled; // [F] taps/F, zero padded
    const cf* wtab;
    cf* Hpos; // [F]
};

template <class Ctx>
AISX_DI void corr_inith_body(Ctx& cx, const CorrInitParams& p)
{
    const int t = cx.tid();
    cf* lds = (cf*)cx.lds();
    cf* ldsX = lds;
    cf* ldsT = lds + 16 * CF_ROW;
    ldsT[t] = p.wtab[(16 * (t >> 3) * (t & 7)) & (CF_F - 1)];
    cx.sync();
    cf x[16], v[2][8];
#pragma unroll
    for (int n1 = 0; n1 < 16; n1++)
        x[n1] = p.taps_scaled[t + CF_T * n1];
    cf_forward(cx, x, p.wtab, ldsX, ldsT, v);
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int k3 = 0; k3 < 8; k3++)
            p.Hpos[(t + CF_T * h) * 8 + k3] = v[h][k3];
}

template <class Ctx>
AISX_DI void corr_main_body(Ctx& cx, const CorrParams& p)
{
    const int t = cx.tid();
    const int c = cx.by();
    const int seg = cx.bx();
    cf* lds = (cf*)cx.lds();
    cf* ldsX = lds;
    cf* ldsT = lds + 16 * CF_ROW;

    const int N = p.N, L = p.L, n = p.n;
    const cf* xin = p.in + (long)c * p.in_stride;
    cf* xout = p.out + (long)c * p.out_stride;
    cf* xcorr = p.corr + (long)c * p.corr_stride;
    const cf* hist = p.hist_in + (long)c * N;
    unsigned long long* abits = p.abits + (long)c * p.abits_stride;

    ldsT[t] = p.wtab[(16 * (t >> 3) * (t & 7)) & (CF_F - 1)];
    cx.sync();
    // value n1 of this thread is window item i = t + 128 n1; in an interior tile it is a
    // correlation output iff i >= N (i < F = N + L always)
    unsigned vmask_int = 0;
#pragma unroll
    for (int n1 = 0; n1 < 16; n1++)
        if (t + CF_T * n1 >= N)
            vmask_int |= 1u << n1;

    for (int tile = 0; tile < p.tiles_per_seg; tile++) {
        const int k0 = (seg * p.tiles_per_seg + tile) * L;
        if (k0 >= n)
            break;
        cf x[16], v[2][8];
        // a tile whose whole window and all L outputs lie inside this call's items (all but the
        // first and the last of a channel): no per-element bounds tests, the few that remain
        // (i < L, i >= N) are decided per 128-element slice in scalar code
        const bool interior = (k0 - N >= 0) && (k0 + L <= n);
        if (interior) {
            const cf* w = xin + (k0 - N);
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++)
                x[n1] = (w + CF_T * n1)[(unsigned)t]; // uniform base + 32-bit lane offset
            // A2: out[k0 + i] = stream[k0 + i - N] = w[i]   (lib/corr_est_cc_impl.cc:184)
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                const int lo = CF_T * n1;
                cf* on = xout + (k0 + lo);
                if (lo + CF_T <= L)
                    on[(unsigned)t] = x[n1];
                else if (lo < L && lo + t < L)
                    on[(unsigned)t] = x[n1];
            }
        } else {
            // window w[i] = stream[k0 - N + i]; stream index < 0 comes from the history
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                const int i = t + CF_T * n1;
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
                const int i = t + CF_T * n1;
                if (i < L && k0 + i < n)
                    xout[k0 + i] = x[n1];
            }
            if (p.corr_hist_zero) {
#pragma unroll
                for (int n1 = 0; n1 < 16; n1++)
                    if (k0 - N + t + CF_T * n1 < 0)
                        x[n1] = mk(0.f, 0.f);
            }
        }
        cf_forward(cx, x, p.wtab, ldsX, ldsT, v);
        // spectrum x H, inverse radix-8
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int q = t + CF_T * h, k1 = q >> 4, k2 = q & 15, swz = (k2 >> 2) & 3;
            const int base = k1 * CF_ROW + k2 * 8;
            const cf* Hq = p.Hpos + q * 8; // spectrum positions (q, 0..7): 64 contiguous bytes, L2 resident
#pragma unroll
            for (int k3 = 0; k3 < 8; k3++)
                v[h][k3] = cmul_fma(v[h][k3], Hq[k3]);
            dft8<true>(cx, v[h]);
#pragma unroll
            for (int pr = 0; pr < 4; pr++) {
                const int ch = base + 2 * (pr ^ swz);
                ldsX[ch] = v[h][2 * pr];
                ldsX[ch + 1] = v[h][2 * pr + 1];
            }
        }
        cx.sync();
        {
            const int k1 = t >> 3, n3 = t & 7;
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) {
                cf a = ldsX[cf_pos(k1, k2 * 8 + n3)];
                x[k2] = (k2 == 0) ? a : cmul_conj_fma(a, ldsT[k2 * 8 + n3]);
            }
            dft16<true>(cx, x);
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++)
                ldsX[cf_pos(k1, n2 * 8 + n3)] = x[n2];
        }
        cx.sync();
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++) {
            cf a = ldsX[cf_pos(k1, t)];
            x[k1] = (k1 == 0) ? a : cmul_conj_fma(a, p.wtab[k1 * t]);
        }
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
                const int m = t + CF_T * n1 - N;
                if (m >= 0 && m < L && k0 + m < n)
                    vmask |= 1u << n1;
            }
        }
        const int kb = k0 + t - N; // output index of value n1: kb + 128 n1
        if (p.dense_corr) {
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++)
                if ((vmask >> n1) & 1u)
                    xcorr[kb + CF_T * n1] = x[n1];
        }
        unsigned hit = 0;
#pragma unroll
        for (int n1 = 0; n1 < 16; n1++) {
            const float mg = mag2(x[n1]);
            hit |= (!(mg <= p.thresh)) ? (1u << n1) : 0u;
        }
        hit &= vmask;
        if (cx.ballot(hit != 0u) != 0ull)
            corr_emit_hits(cx, p, hit, vmask, x, xcorr, abits, kb, CF_T);
        cx.sync();
    }
    // carry the last N stream samples to the next call (set_history(N+1), :95)
    if (seg == p.nseg - 1) {
        cf* ho = p.hist_out + (long)c * N;
        for (int j = t; j < N; j += CF_T) {
            const int s = n - N + j;
            ho[j] = (s >= 0) ? xin[s] : hist[N + s];
        }
    }
}

// ---------------------------------------------------------------------------
// A5: peak search and tag emission, one workgroup per channel.
// ---------------------------------------------------------------------------
struct ResolveParams {
    const unsigned long long* abits; long abits_stride;
    const cf* corr; long corr_stride; int dense_corr;
    int L; // outputs per tile of the main kernel (its FFT size is L + N)
    const cf* in; long in_stride;
    const cf* hist_in; // history the correlation of this call started from
    int corr_hist_zero; // ... or zeros (first call after set_symbols, see CorrParams)
    const cf* taps;    // d_symbols as stored (reversed conjugate), N entries
    int n, N, isps;
    unsigned mark_delay;
    unsigned long long written; // nitems_written(0)
    int emit_port1;
    tag_rec* tags; int tag_cap; int* tag_count;
    const float* atan_tab;
};

template <class Ctx>
AISX_DI void resolve_direct_mag2(Ctx& cx, const ResolveParams& p, int c, int pk, bool want0, bool want2, float& m0, float& m2)
{
    // corr[k] = sum_j taps[j] * x[k - j], recomputed in direct form (double
    // accumulation) for the below-threshold neighbours k = pk - 1 / pk + 1 of a peak; only
    // feeds the 3-point centre of mass.  Both neighbours in one pass over the taps (one
    // memory round trip, not two).
    const int lane = cx.tid() & 63;
    const cf* xin = p.in + (long)c * p.in_stride;
    const cf* hist = p.hist_in + (long)c * p.N;
    double ar0 = 0.0, ai0 = 0.0, ar2 = 0.0, ai2 = 0.0;
    // (no branch inside the loops: the loads of all trips go out together)
    auto pass = [&](auto w0, auto w2) {
        constexpr bool W0 = decltype(w0)::value, W2 = decltype(w2)::value;
        for (int j = lane; j < p.N; j += 64) {
            const int s0 = pk - 1 - j, s2 = pk + 1 - j;
            const cf tv = p.taps[j];
            if (W0) {
                cf xv = (s0 >= 0) ? xin[s0] : hist[p.N + s0];
                if (s0 < 0 && p.corr_hist_zero)
                    xv = mk(0.f, 0.f);
                ar0 += (double)tv.re * (double)xv.re - (double)tv.im * (double)xv.im;
                ai0 += (double)tv.re * (double)xv.im + (double)tv.im * (double)xv.re;
            }
            if (W2) {
                cf xv = (s2 >= 0) ? xin[s2] : hist[p.N + s2];
                if (s2 < 0 && p.corr_hist_zero)
                    xv = mk(0.f, 0.f);
                ar2 += (double)tv.re * (double)xv.re - (double)tv.im * (double)xv.im;
                ai2 += (double)tv.re * (double)xv.im + (double)tv.im * (double)xv.re;
            }
        }
    };
    typedef std::true_type yes;
    typedef std::false_type no;
    if (want0 && want2)
        pass(yes{}, yes{});
    else if (want0)
        pass(yes{}, no{});
    else
        pass(no{}, yes{});
    if (want0) { // (wave-uniform)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            ar0 += cx.shfl_xor_f64(ar0, o);
            ai0 += cx.shfl_xor_f64(ai0, o);
        }
        m0 = mag2(mk((float)ar0, (float)ai0));
    }
    if (want2) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            ar2 += cx.shfl_xor_f64(ar2, o);
            ai2 += cx.shfl_xor_f64(ai2, o);
        }
        m2 = mag2(mk((float)ar2, (float)ai2));
    }
}

// The greedy scan of lib/corr_est_cc_impl.cc:197-270 over the hit bitmask words [base0, base1) of channel c (base0 a
// multiple of 64), starting at item i0: first hit at or behind i, climb to the local maximum, centre of mass, phase,
// emit(pk, mag^2, phase, centre) -- wave-uniform arguments --, on at pk + isps.  One wave; `lane` is the lane in it.
template <class Ctx, class Emit>
AISX_DI void resolve_scan(Ctx& cx, const ResolveParams& p, int c, const float* atab, int base0, int base1, int i0, Emit&& emit)
{
    const int lane = cx.tid() & 63;
    const unsigned long long* A = p.abits + (long)c * p.abits_stride;
    const cf* corr = p.corr + (long)c * p.corr_stride;
    const int n = p.n;
    const int nwords = (n + 63) >> 6;
    int i = i0;
    // the window in hand: items wq0 .. wq0 + 63 (lane j: value, mag^2), the lanes whose value is in
    // hand, the lanes from which the climb steps on
    bool win = false;
    int wq0 = 0;
    cf wcv = mk(0.f, 0.f);
    float wmg = 0.f;
    unsigned long long wHV = 0ull, wC = 0ull;
    for (int base = base0; base < base1; base += 64) {
        if ((long)(base + 64) * 64 <= (long)i)
            continue;
        const int wi = base + lane;
        const unsigned long long w = (wi < nwords) ? A[wi] : 0ull;
        for (;;) {
            unsigned long long we = w;
            const long lo = (long)wi * 64;
            if (lo + 64 <= (long)i)
                we = 0ull;
            else if (lo < (long)i)
                we &= (~0ull) << (i - lo);
            const unsigned long long nz = cx.ballot(we != 0ull);
            if (!nz)
                break;
            const int src = cx.ctz64(nz);
            const unsigned long long word = cx.shfl_u64(we, src);
            int pk = (base + src) * 64 + cx.ctz64(word);
            // One window per burst: lane j looks at item q0 + j -- its above-threshold bit and its
            // correlation value, fetched in one go -- so that the climb (:202-204) and the centre of
            // mass (:219-227) cost one memory round trip instead of one per step; and the next
            // detections of the same burst (isps items on, :270: usually still inside the window)
            // none at all.  A fresh window starts at q0 = pk - 1.
            cf cp = mk(0.f, 0.f);
            float mp = 0.f, m0 = 0.f, m2 = 0.f;
            bool have0 = false, have2 = false;
            for (;;) {
                int j = pk - wq0; // lane of item pk in the window in hand
                if (!(win && j >= 1 && j <= 61)) {
                    const int q0 = pk - 1;
                    const int wa = (q0 < 0 ? 0 : q0) >> 6;
                    // (words base .. base + 63 of the bitmask are in the wave's registers)
                    const int ra = wa - base, rb = wa + 1 - base;
                    unsigned long long WA = cx.shfl_u64(w, ra & 63), WB = cx.shfl_u64(w, rb & 63);
                    if (ra < 0 || ra > 63)
                        WA = A[wa];
                    if (rb < 0 || rb > 63)
                        WB = (wa + 1 < nwords) ? A[wa + 1] : 0ull;
                    const int pos = q0 + lane;
                    const bool inside = pos >= 0 && pos < n;
                    const unsigned long long wsel = ((pos >> 6) == wa) ? WA : WB;
                    const bool above = inside && ((pos >> 6) <= wa + 1) && ((wsel >> (pos & 63)) & 1ull);
                    const unsigned long long AB = cx.ballot(above);
                    // below-threshold items the main kernel left in the scratch next to a hit
                    // (corr_emit_hits): element m of its tile, same tile and same wave as the hit
                    bool nbr = false;
                    if (!p.dense_corr) {
                        const int F = p.L + p.N;
                        const int t0 = ((q0 < 0 ? 0 : q0) / p.L) * p.L; // first output of the tile the window starts in
                        int m = pos - t0 + p.N;
                        if (m >= F)
                            m -= p.L;
                        const bool left_hit = lane > 0 && ((AB >> (lane - 1)) & 1ull);   // item pos - 1 is a hit
                        const bool right_hit = lane < 63 && ((AB >> (lane + 1)) & 1ull); // item pos + 1 is a hit
                        nbr = inside && !above &&
                              ((left_hit && m - 1 >= p.N && ((m - 1) & 63) != 63) || (right_hit && m + 1 < F && ((m + 1) & 63) != 0));
                    }
                    wcv = mk(0.f, 0.f);
                    if (inside && (above || nbr || p.dense_corr))
                        wcv = corr[pos];
                    wmg = mag2(wcv);
                    // climb: from item q to q + 1 while q + 1 is above threshold and larger
                    const float mg_next = cx.shfl_down_f32(wmg, 1);
                    wHV = AB | cx.ballot(nbr); // lanes whose value is in hand
                    const bool step_ok = (lane < 63) && ((AB >> (lane + 1)) & 1ull) && (wmg < mg_next);
                    wC = cx.ballot(step_ok);
                    wq0 = q0;
                    win = true;
                    j = 1;
                }
                const int run = cx.ctz64(~(wC >> j)); // consecutive climbs from lane j (= pk)
                const int jp = j + run;               // lane of the local maximum, if it is inside the window
                if (jp >= 63) { // the climb runs off the window: move the window there and go on
                    pk = wq0 + 62;
                    win = false;
                    continue;
                }
                pk = wq0 + jp;
                cp = mk(cx.shfl_f32(wcv.re, jp), cx.shfl_f32(wcv.im, jp));
                mp = mag2(cp);
                have0 = p.dense_corr || ((wHV >> (jp - 1)) & 1ull);
                have2 = p.dense_corr || ((wHV >> (jp + 1)) & 1ull);
                m0 = cx.shfl_f32(wmg, jp - 1);
                m2 = cx.shfl_f32(wmg, jp + 1);
                break;
            }
            // centre of mass (:219-227)
            double center = 0.0;
            if (pk > 0 && pk < n - 1) {
                if (!have0 || !have2)
                    resolve_direct_mag2(cx, p, c, pk, !have0, !have2, m0, m2);
                double nom = 0, den = 0;
                nom += (double)(1.0f * m0);
                den += (double)m0;
                nom += (double)(2.0f * mp);
                den += (double)mp;
                nom += (double)(3.0f * m2);
                den += (double)m2;
                center = nom / den - 2.0;
            }
            const float phase = fast_atan2f_tab(cp.im, cp.re, atab); // :247
            emit(pk, mp, phase, center);
            i = pk + p.isps; // :270
            if ((long)i >= (long)(base + 64) * 64)
                break;
        }
    }
}

// tags of one detection, the d-th of its channel (:247-262): written while they fit, counted always
AISX_DI void resolve_write_tags(const ResolveParams& p, tag_rec* tags, int c, int d, int pk, float mp, float phase, double center)
{
    const int per = p.emit_port1 ? 7 : 4;
    const int ntag = d * per;
    const unsigned long long o0 = p.written + (unsigned long long)pk;
    const unsigned long long o1 = o0 + p.mark_delay;
    if (ntag + 4 <= p.tag_cap) {
        tags[ntag + 0] = tag_rec{ o0, (double)mp, KEY_CORR_START, c };
        tags[ntag + 1] = tag_rec{ o1, (double)phase, KEY_PHASE_EST, c };
        tags[ntag + 2] = tag_rec{ o1, center, KEY_TIME_EST, c };
        tags[ntag + 3] = tag_rec{ o1, (double)mp, KEY_CORR_EST, c };
    }
    if (p.emit_port1 && ntag + 7 <= p.tag_cap) {
        tags[ntag + 4] = tag_rec{ o0, (double)phase, KEY_PHASE_EST | 0x100, c };
        tags[ntag + 5] = tag_rec{ o0, center, KEY_TIME_EST | 0x100, c };
        tags[ntag + 6] = tag_rec{ o0, (double)mp, KEY_CORR_EST | 0x100, c };
    }
}

// One workgroup of RSV_WAVES waves per channel (round 6; one wave per channel before: 0.36 / 0.58 ms per call at 4096 /
// 8192 channels, a chain of dependent memory round trips per detection, ~55 detections one after the other).
// The scan is sequential only through its cursor (i = pk + isps behind a detection), and the cursor cannot carry anything
// across a gap: if the isps items in front of item s hold no hit, no climb reaches s (a climb walks over hits only) and
// every detection in front of s leaves the cursor at or below s -- the scan from s on is the scan that starts at s.
// So the bitmask is cut into blocks of 64 words (4096 items); a block whose start is such a gap ("clean": the last
// isps bits of the word in front of it are zero; block 0 always) opens a REGION that runs to the next clean block,
// and the regions of a channel are scanned by different waves at the same time.  Detections wait in LDS; behind a
// barrier each wave knows how many detections precede its regions and writes its tags where the sequential scan
// would have put them.  More detections in one wave than its share of the LDS holds, more than RSV_MAXB blocks, isps > 64: the
// workgroup's first wave runs the sequential scan instead (same tags).
#ifndef RSV_WAVES_MAX
#define RSV_WAVES_MAX 16
#endif
constexpr int RSV_WAVES = RSV_WAVES_MAX;
constexpr int RSV_DET_WAVE = 128; // detections a wave can hold in LDS (64 with sixteen waves: 24 KB per workgroup at most)
AISX_HD int rsv_det_cap(int nwv) { return nwv >= 16 ? 64 : RSV_DET_WAVE; }
constexpr int RSV_MAXB = 256;
constexpr int RSV_ATAB_BYTES = 1056; // 257 floats, padded to 32 bytes
struct rsv_det {
    double center;
    int pk;
    float mp, phase;
    int pad;
};
// LDS of a workgroup of nwv waves (one wave: the sequential scan, nothing but the table)
AISX_HD int rsv_lds_bytes(int nwv) { return RSV_ATAB_BYTES + (nwv > 1 ? (RSV_MAXB + 8) * 4 + nwv * rsv_det_cap(nwv) * (int)sizeof(rsv_det) : 16); }
// Waves per channel for a launch over nchan channels: what fills the chip's wave slots (8192 on 256 CUs), no more --
// with a wave on every slot already (8192 channels) more waves per channel only add idle ones (measured: 0.99 against
// 0.58 ms per call with sixteen), with few channels the scan's chain of round trips is cut sixteen-fold.
AISX_HD int rsv_waves_for(int nchan)
{
    int w = RSV_WAVES;
    while (w > 1 && (long)w * nchan > 8192)
        w >>= 1;
    return w;
}

template <class Ctx>
AISX_DI void corr_resolve_body(Ctx& cx, const ResolveParams& p)
{
    const int tid = cx.tid(), lane = tid & 63, wv = tid >> 6, nwv = cx.nthreads() >> 6;
    const int c = cx.bx();
    const unsigned long long* A = p.abits + (long)c * p.abits_stride;
    tag_rec* tags = p.tags + (long)c * p.tag_cap;
    const int nwords = (p.n + 63) >> 6;
    const int nblk = (nwords + 63) >> 6;
    // fast_atan2f's table next to the waves (a detection is a chain of dependent memory round
    // trips; this one becomes an LDS read)
    float* atab = (float*)cx.lds();
    int* cnt = (int*)(cx.lds() + RSV_ATAB_BYTES); // detections per region, by the block that opens it
    int* ovf = cnt + RSV_MAXB;
    const int det_cap = rsv_det_cap(nwv);
    rsv_det* det = (rsv_det*)(cx.lds() + RSV_ATAB_BYTES + (RSV_MAXB + 8) * 4) + wv * det_cap;
    const bool par = nwv > 1 && nblk > 1 && nblk <= RSV_MAXB && p.isps >= 1 && p.isps <= 64;
    for (int k = tid; k < 257; k += cx.nthreads())
        atab[k] = p.atan_tab[k];
    if (nwv > 1)
        for (int k = tid; k < RSV_MAXB + 8; k += cx.nthreads())
            cnt[k] = 0;
    cx.sync();
    auto clean = [&](int b) { // (wave-uniform)
        if (b == 0)
            return true;
        return (A[64 * b - 1] >> (64 - p.isps)) == 0ull;
    };
    int mine = 0; // detections this wave holds
    if (par) {
        for (int b = wv; b < nblk; b += nwv) {
            if (!clean(b))
                continue;
            int e = b + 1;
            while (e < nblk && !clean(e))
                e++;
            int here = 0;
            resolve_scan(cx, p, c, atab, 64 * b, e < nblk ? 64 * e : nwords, 4096 * b, [&](int pk, float mp, float phase, double center) {
                if (lane == 0 && mine < det_cap)
                    det[mine] = rsv_det{ center, pk, mp, phase, 0 };
                mine++;
                here++;
            });
            if (lane == 0) {
                cnt[b] = here;
                if (mine > det_cap)
                    *ovf = 1;
            }
        }
    }
    cx.sync();
    if (!par || (nwv > 1 && *ovf != 0)) {
        if (wv == 0) {
            int nd = 0;
            resolve_scan(cx, p, c, atab, 0, nwords, 0, [&](int pk, float mp, float phase, double center) {
                if (lane == 0)
                    resolve_write_tags(p, tags, c, nd, pk, mp, phase, center);
                nd++;
            });
            if (lane == 0)
                p.tag_count[c] = nd * (p.emit_port1 ? 7 : 4);
        }
        return;
    }
    // detections in front of each of this wave's regions: the counts of all earlier blocks
    int beg = 0;
    for (int b = wv; b < nblk; b += nwv) {
        const int here = cnt[b];
        if (here == 0)
            continue;
        int before = 0;
        for (int k = 0; k < b; k++)
            before += cnt[k];
        for (int k = lane; k < here; k += 64) {
            const rsv_det d = det[beg + k];
            resolve_write_tags(p, tags, c, before + k, d.pk, d.mp, d.phase, d.center);
        }
        beg += here;
    }
    if (tid == 0) {
        int total = 0;
        for (int k = 0; k < nblk; k++)
            total += cnt[k];
        p.tag_count[c] = total * (p.emit_port1 ? 7 : 4);
    }
}

} // namespace aisx
