// k_corr4e.h -- the F = 4096 correlator of k_corr4d.h (same contract, same two window images, same
// LDS-DMA prefetch) on TWICE the waves: 512 threads x 8 points, four radix-8 register passes per
// direction, at most 128 VGPRs -- four waves per SIMD with two workgroups per CU.
//
// Why (profiles/r05_corr_main_pmc.json, k_corr4d_main<896>): at 256 VGPRs and 71 680 B of LDS a
// SIMD holds two waves (one of each resident workgroup); a wave executes 43 % of its life, waits
// at s_waitcnt / barriers 30 %, is stalled at issue 27 %.  0.60 ms of VALU time, ~0.35 ms of LDS
// time and 0.78 ms of HBM time per launch have to overlap, and two waves per SIMD cannot do it.
// A third WORKGROUP per CU does not fit (the second image is what the LDS goes to); more WAVES on
// the same footprint do: 4096 = 8 x 8 x 8 x 8, a thread holds 8 complex points (+ its 8 values of
// H, + 3 x 7 twiddles), the transform takes one more LDS exchange per direction than 16 x 16 x 16.
//
// Index plan (forward, decimation in frequency; the inverse mirrors it, decimation in time):
//   n = 512 n1 + 64 n2 + 8 n3 + n4          k = k1 + 8 k2 + 64 k3 + 512 k4
//   pass 1  thread t = 64 n2 + 8 n3 + n4    x[n1] -> [k1], times W_4096^{k1 t}
//   pass 2  thread (k1; r2 = 8 n3 + n4)     x[n2] -> [k2], times W_512^{k2 r2}
//   pass 3  thread (k1; k2, n4)             x[n3] -> [k3], times W_64^{k3 n4}
//   pass 4  thread (k1; k2, k3)             x[n4] -> [k4]            spectrum position q = t, value k4
// k1 is the wave in passes 2-4: only the exchange between pass 1 and pass 2 crosses waves (a
// workgroup barrier); the other two are exchanges among the lanes of ONE wave (wave_lds_sync).
// Three barriers per tile, as k_corr4d.h.
//
// LDS image (8 rows x 576 complex = 36 864 B; two images + nothing else = 73 728 B, two
// workgroups per CU):
//   * a row is eight blocks of 64 items at a pitch of 72 (pass 3 reads with lanes (k2, n4) at
//     stride 8 n3: four blocks' worth of lanes per 32-lane group -> pitch 72 = 8 mod 32 keeps them on
//     distinct banks);
//   * inside an 8-item group (k3) the 16-byte chunk index is XOR-ed with (k3 >> 1) ^ (k2 & 3):
//     pass 4 moves whole groups as 4 x ds_read_b128 / ds_write_b128 per lane, conflict free on
//     both sides (the read's lane groups are {0-3, 12-15, 20-27}, ...; the write's are 8 contiguous lanes);
//   * the window arrives in natural order, 128 items (one DMA wave-instruction, 1 KiB) at a pitch
//     of 144: the first pass reads it there and writes its result in the transform's layout IN
//     PLACE -- wave w reads columns [144 (w >> 1) + 64 (w & 1), + 64) and writes into
//     [144 (w >> 1) + 72 (w & 1), + 64): its own columns and padding nobody reads.
#pragma once
#include <type_traits>
#include "k_corr4d.h"

namespace aisx {

constexpr int CE_T = 512;            // threads per transform (8 waves)
constexpr int CE_BLK = 72;           // pitch of a 64-item block
constexpr int CE_ROW = 8 * CE_BLK;   // row pitch in complex elements
constexpr int CE_IMG = 8 * CE_ROW;   // complex slots per window image
// second / third pass twiddles: 1 = in registers (14 VGPRs each), 0 = a table in LDS (7 x 64 / 7 x 8 items)
#ifndef CE_W2_REGS
#define CE_W2_REGS 1
#endif
#ifndef CE_W3_REGS
#define CE_W3_REGS 0
#endif
// the exchange between the second and the third pass (register index against lane bits 3 .. 5 of ONE wave): 1 = in
// registers (Ctx::xpose8_lane_hi: permlane swaps + DPP moves), 0 = through the window image in LDS
#ifndef CE_XP2
#define CE_XP2 0
#endif
// first / last pass twiddles W_4096^{k t}, k = 1 .. 7: 1 = all seven in registers (14 VGPRs),
// 0 = the powers 1, 2, 4 (6 VGPRs) and four more products per pass
#ifndef CE_W1_REGS
#define CE_W1_REGS 1
#endif
constexpr int CE_TAB = (CE_W2_REGS ? 0 : 7 * 64) + (CE_W3_REGS ? 0 : 7 * 8);
constexpr int CE_LDS_ELEMS = 2 * CE_IMG + CE_TAB;
constexpr int CE_LDS_BYTES = CE_LDS_ELEMS * 8; // 73 728 (77 760 with the tables): two workgroups per CU

// natural-order slot of window item i (what the DMA writes and the first pass reads)
AISX_HD int ce_nat(int i) { return (i >> 9) * CE_ROW + ((i >> 7) & 3) * 144 + (i & 127); }
// transform-order column of (block b, item r of the block), b = 0 .. 7, r = 8 g + e
AISX_HD int ce_col(int b, int r)
{
    const int g = r >> 3, e = r & 7;
    return b * CE_BLK + g * 8 + ((((e >> 1) ^ (g >> 1) ^ (b & 3)) << 1) | (e & 1));
}

template <class Ctx, int NV>
AISX_DI void corr_emit_hits_n(Ctx& cx, const CorrParams& p, unsigned hit, unsigned vmask, const cf (&x)[NV], cf* xcorr,
                              unsigned long long* abits, int kb, int stride)
{
    // (k_corr.h: corr_emit_hits, for NV values per thread)
    const unsigned nb = (cx.lane_prev_u32(hit) | cx.lane_next_u32(hit)) & vmask & ~hit;
#pragma unroll
    for (int n1 = 0; n1 < NV; n1++) {
        if (((hit | nb) >> n1) & 1u) {
            const int k = kb + stride * n1;
            if (!p.dense_corr)
                xcorr[k] = x[n1];
            if ((hit >> n1) & 1u)
                cx.atomic_or64(&abits[k >> 6], 1ull << (k & 63));
        }
    }
}

// The per-thread constants of the 8 x 8 x 8 x 8 plan and its passes over one image.  W2R / W3R: the
// second / third pass twiddles in registers, or in tables in LDS (W_512 first, then W_64) at `ldsT`.
template <class Ctx, bool W2R = (CE_W2_REGS != 0), bool W3R = (CE_W3_REGS != 0)>
struct Ce {
    Ctx& cx;
    int t, lane, k1;
    // column offsets (complex slots, without the row): pass 1 reads c_nat (+ 576 n1) and writes
    // c_p1 (+ 576 k1); pass 2 touches c_p2[b & 3] + 72 b; pass 3 c_p3[k3 >> 1] + 8 k3; pass 4 the
    // chunks c_p4 + 2 (pr ^ s4)
    int c_nat, c_p1, c_p2[4], c_p3[4], c_p4, s4;
    cf w1[8], w2[8], w3[8];
    const cf *T2, *T3;

    AISX_DI Ce(Ctx& c, const cf* wtab, cf* ldsT) : cx(c)
    {
        t = cx.tid();
        lane = t & 63;
        k1 = t >> 6;
        c_nat = (t >> 7) * 144 + (t & 127);
        c_p1 = ce_col(t >> 6, lane);
#pragma unroll
        for (int v = 0; v < 4; v++) {
            // pass 2: item r2 = lane of block b, b & 3 = v
            c_p2[v] = k1 * CE_ROW + ce_col(v, lane) - v * CE_BLK;
            // pass 3: lane = (k2, n4), group k3 with k3 >> 1 = v
            const int k2 = lane >> 3, n4 = lane & 7;
            c_p3[v] = k1 * CE_ROW + ce_col(k2, 16 * v + n4) - 16 * v;
        }
        {
            const int k2 = lane >> 3, k3 = lane & 7;
            c_p4 = k1 * CE_ROW + k2 * CE_BLK + k3 * 8;
            s4 = (k3 >> 1) ^ (k2 & 3);
        }
        w1[0] = mk(1.f, 0.f);
#if CE_W1_REGS
#pragma unroll
        for (int k = 1; k < 8; k++)
            w1[k] = wtab[(k * t) & (CF4_F - 1)];
#else
        w1[1] = wtab[t];
        w1[2] = wtab[(2 * t) & (CF4_F - 1)];
        w1[4] = wtab[(4 * t) & (CF4_F - 1)];
#endif
        // (tables behind the two images: W_512^{k r}, k = 1 .. 7, r = 0 .. 63, then W_64^{k e}, e = 0 .. 7;
        // the caller puts a barrier behind this)
        T2 = T3 = ldsT;
        if constexpr (W2R) {
            w2[0] = mk(1.f, 0.f);
#pragma unroll
            for (int k = 1; k < 8; k++)
                w2[k] = wtab[(8 * k * lane) & (CF4_F - 1)]; // W_512^{k2 r2}
        } else {
            if (t < 7 * 64)
                ldsT[t] = wtab[(8 * (t / 64 + 1) * (t & 63)) & (CF4_F - 1)];
            T2 = ldsT + lane - 64;
        }
        if constexpr (W3R) {
            w3[0] = mk(1.f, 0.f);
#pragma unroll
            for (int k = 1; k < 8; k++)
                w3[k] = wtab[(64 * k * (lane & 7)) & (CF4_F - 1)]; // W_64^{k3 n4}
        } else {
            cf* const ldsT3 = ldsT + (W2R ? 0 : 7 * 64);
            if (t < 7 * 8)
                ldsT3[t] = wtab[(64 * (t / 8 + 1) * (t & 7)) & (CF4_F - 1)];
            T3 = ldsT3 + (lane & 7) - 8;
        }
    }
    AISX_DI cf tw2(int k) const
    {
        if constexpr (W2R)
            return w2[k];
        else
            return ld8(T2 + 64 * k);
    }
    AISX_DI cf tw3(int k) const
    {
        if constexpr (W3R)
            return w3[k];
        else
            return ld8(T3 + 8 * k);
    }
    // x[k] *= W^{k t} (conjugated if INV), k = 1 .. 7
    template <bool INV>
    AISX_DI void twiddle1(cf (&x)[8]) const
    {
#if CE_W1_REGS
#pragma unroll
        for (int k = 1; k < 8; k++)
            x[k] = INV ? cx.cmul_conj(x[k], w1[k]) : cx.cmul(x[k], w1[k]);
#else
        const cf a = w1[1], b = w1[2], c = w1[4];
        const cf w3_ = cx.cmul(a, b), w5 = cx.cmul(c, a), w6 = cx.cmul(c, b), w7 = cx.cmul(c, w3_);
#define AISX_AP(k, w) x[k] = INV ? cx.cmul_conj(x[k], (w)) : cx.cmul(x[k], (w))
        AISX_AP(1, a);
        AISX_AP(2, b);
        AISX_AP(3, w3_);
        AISX_AP(4, c);
        AISX_AP(5, w5);
        AISX_AP(6, w6);
        AISX_AP(7, w7);
#undef AISX_AP
#endif
    }

    // forward pass 1 on x[n1] = window item t + 512 n1 (already in registers); result to image A
    AISX_DI void fwd1(cf* A, cf (&x)[8]) const
    {
        dft8<false>(cx, x);
        twiddle1<false>(x);
#pragma unroll
        for (int k = 0; k < 8; k++)
            st8(A + k * CE_ROW + c_p1, x[k]);
    }
    // forward passes 2, 3, 4 (behind the workgroup barrier that follows fwd1); x[k4] = spectrum
    // position (q = t, k4) on exit
    AISX_DI void fwd234(cf* A, cf (&x)[8]) const
    {
#pragma unroll
        for (int b = 0; b < 8; b++)
            x[b] = ld8(A + c_p2[b & 3] + b * CE_BLK);
        cx.wave_sync();
        dft8<false>(cx, x);
#pragma unroll
        for (int b = 1; b < 8; b++)
            x[b] = cx.cmul(x[b], tw2(b));
#if CE_XP2
        cx.xpose8_lane_hi(x); // (k2 in the registers, n3 in lane bits 3 .. 5) -> (n3 in the registers, k2 in the lane)
#else
#pragma unroll
        for (int b = 0; b < 8; b++)
            st8(A + c_p2[b & 3] + b * CE_BLK, x[b]);
        cx.wave_lds_sync();
#pragma unroll
        for (int g = 0; g < 8; g++)
            x[g] = ld8(A + c_p3[g >> 1] + g * 8);
        cx.wave_sync();
#endif
        dft8<false>(cx, x);
#pragma unroll
        for (int g = 1; g < 8; g++)
            x[g] = cx.cmul(x[g], tw3(g));
#pragma unroll
        for (int g = 0; g < 8; g++)
            st8(A + c_p3[g >> 1] + g * 8, x[g]);
        cx.wave_lds_sync();
#pragma unroll
        for (int pr = 0; pr < 4; pr++)
            ld16(A + c_p4 + 2 * (pr ^ s4), x[2 * pr], x[2 * pr + 1]);
        cx.wave_sync();
        dft8<false>(cx, x);
    }
    // inverse passes 4, 3, 2 on x[k4] (spectrum x H); the caller puts a workgroup barrier behind it
    AISX_DI void inv432(cf* A, cf (&x)[8]) const
    {
        dft8<true>(cx, x);
#pragma unroll
        for (int pr = 0; pr < 4; pr++)
            st16(A + c_p4 + 2 * (pr ^ s4), x[2 * pr], x[2 * pr + 1]);
        cx.wave_lds_sync();
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const cf a = ld8(A + c_p3[g >> 1] + g * 8);
            x[g] = (g == 0) ? a : cx.cmul_conj(a, tw3(g));
        }
        cx.wave_sync();
        dft8<true>(cx, x);
#if CE_XP2
        cx.xpose8_lane_hi(x);
#pragma unroll
        for (int b = 1; b < 8; b++)
            x[b] = cx.cmul_conj(x[b], tw2(b));
#else
#pragma unroll
        for (int g = 0; g < 8; g++)
            st8(A + c_p3[g >> 1] + g * 8, x[g]);
        cx.wave_lds_sync();
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const cf a = ld8(A + c_p2[b & 3] + b * CE_BLK);
            x[b] = (b == 0) ? a : cx.cmul_conj(a, tw2(b));
        }
        cx.wave_sync();
#endif
        dft8<true>(cx, x);
#pragma unroll
        for (int b = 0; b < 8; b++)
            st8(A + c_p2[b & 3] + b * CE_BLK, x[b]);
    }
    // inverse pass 1: x[n1] = y[t + 512 n1]
    AISX_DI void inv1(const cf* A, cf (&x)[8]) const
    {
#pragma unroll
        for (int k = 0; k < 8; k++)
            x[k] = ld8(A + k * CE_ROW + c_p1);
        twiddle1<true>(x);
        dft8<true>(cx, x);
    }
};

// template spectrum in the plan's own position order: Hpos[8 t + k4]
template <class Ctx>
AISX_DI void corr4e_inith_body(Ctx& cx, const CorrInitParams& p)
{
    cf* lds = (cf*)cx.lds();
    Ce<Ctx> ce(cx, p.wtab, lds + 2 * CE_IMG);
    const int t = ce.t;
    cx.sync();
    cf x[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++)
        x[n1] = p.taps_scaled[t + CE_T * n1];
    ce.fwd1(lds, x);
    cx.sync();
    ce.fwd234(lds, x);
#pragma unroll
    for (int k4 = 0; k4 < 8; k4++)
        p.Hpos[t * 8 + k4] = x[k4];
}

// -DCE_PROF (experiments): every wave adds up the constant-rate clock (100 MHz) over the sections of its tile
// loop -- 0 wait for the window, 1 barrier, 2 window to registers / pass-through / next window, 3 pass 1,
// 4 barrier, 5 passes 2-4, H, inverse 4-2, 6 barrier, 7 last pass and threshold -- into g_ce_prof[]
#if defined(CE_PROF) && defined(__HIP_DEVICE_COMPILE__)
extern __device__ unsigned long long g_ce_prof[16];
#define CE_TICK(i)                                                     \
    do {                                                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
        prof_acc[i] += now_ - prof_t;                                  \
        prof_t = now_;                                                 \
    } while (0)
#else
#define CE_TICK(i) do { } while (0)
#endif

// -DCE_DBG=mask (experiments, timing only -- the results are wrong): leave out 1 the pass-through stores, 2 the
// DMA of the next window, 4 the workgroup barriers of the tile loop, 8 the threshold test, 16 passes 2-4 / H /
// inverse 4-2, 32 the wait for the window, 64 first and last pass
#ifndef CE_DBG
#define CE_DBG 0
#endif
// NC: the template length as a compile-time constant, or 0 for the run-time version (k_corr4d.h)
template <class Ctx, int NC>
AISX_DI void corr4e_main_body(Ctx& cx, const CorrParams& p)
{
    cf* lds = (cf*)cx.lds();
    Ce<Ctx> ce(cx, p.wtab, lds + 2 * CE_IMG);
    const int t = ce.t, lane = ce.lane;
    const int wave = cx.wave_id();
    const int c = cx.by();
    const int seg = cx.bx();

    const int N = NC ? NC : p.N, L = CF4_F - N, n = p.n;
    const cf* xin = p.in + (long)c * p.in_stride;
    cf* xout = p.out + (long)c * p.out_stride;
    cf* xcorr = p.corr + (long)c * p.corr_stride;
    const cf* hist = p.hist_in + (long)c * N;
    unsigned long long* abits = p.abits + (long)c * p.abits_stride;

    cf H[8];
#pragma unroll
    for (int k4 = 0; k4 < 8; k4++)
        H[k4] = p.Hpos[t * 8 + k4];

    unsigned vmask_int = 0; // value n1 of this thread is window item t + 512 n1: an output iff >= N
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++)
        if (t + CE_T * n1 >= N)
            vmask_int |= 1u << n1;

    const auto bin = cx.make_buf(xin, (unsigned)n * 8u);
    const auto bout = cx.make_buf(xout, (unsigned)n * 8u);
    const int P0 = N / CD_PIECE; // first DMA piece that holds new items
    const unsigned lds0 = cx.lds_addr(lds);

    // the new items of the window of the tile whose outputs start at k0: pieces P0 .. 31 over the
    // eight waves; item i of the window is stream item k0 - N + i
#if (CE_DBG & 128) && defined(__HIP_DEVICE_COMPILE__)
    // (timing only: the same bytes by plain loads into registers nobody reads -- no LDS writes)
    typedef float dbg_f4 __attribute__((ext_vector_type(4)));
    dbg_f4 dbg_land[4];
#endif
    auto issue_window = [&](unsigned img, int k0) {
        if (CE_DBG & 2)
            return;
#if (CE_DBG & 128) && defined(__HIP_DEVICE_COMPILE__)
        int q = 0;
        for (int pc = P0 + wave; pc < CF4_F / CD_PIECE; pc += 8, q++) {
            const unsigned off = (unsigned)(k0 - N + CD_PIECE * pc + 2 * lane) * 8u;
            typedef int v4i_ __attribute__((ext_vector_type(4)));
            v4i_ w;
            w.x = (int)(unsigned)(size_t)bin.base;
            w.y = (int)(unsigned)(((size_t)bin.base >> 32) & 0xffffu);
            w.z = (int)bin.nbytes;
            w.w = 0x00020000;
            if (q == 0) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen nt" : "=v"(dbg_land[0]) : "v"(off), "s"(w) : "memory");
            if (q == 1) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen nt" : "=v"(dbg_land[1]) : "v"(off), "s"(w) : "memory");
            if (q == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen nt" : "=v"(dbg_land[2]) : "v"(off), "s"(w) : "memory");
            if (q == 3) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen nt" : "=v"(dbg_land[3]) : "v"(off), "s"(w) : "memory");
        }
        return;
#endif
        for (int pc = P0 + wave; pc < CF4_F / CD_PIECE; pc += 8) {
            const unsigned dst = img + (unsigned)(((pc >> 2) * CE_ROW + (pc & 3) * 144) * 8);
            const unsigned off = (unsigned)(k0 - N + CD_PIECE * pc + 2 * lane) * 8u;
            cx.dma16(bin, off, dst);
        }
    };

    const int tile0 = seg * p.tiles_per_seg;
    int ntile = p.tiles_per_seg;
    {
        const int left = (n - tile0 * L + L - 1) / L;
        ntile = ntile < left ? ntile : left;
    }

    // ---- first window of the segment (k_corr4d.h): new items by DMA, the N before them by plain loads
    if (ntile > 0) {
        const int k0 = tile0 * L;
        issue_window(lds0, k0);
        const int Npro = N + (N & 1);
        cf pro[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int i = t + CE_T * m, s = k0 - N + i;
            pro[m] = mk(0.f, 0.f);
            if (i < Npro)
                pro[m] = (s < 0) ? hist[N + s] : ((s < n) ? xin[s] : mk(0.f, 0.f));
        }
        cx.wait_dma();
        cx.lds_barrier(); // (the tables of Ce, if any, are behind this barrier too)
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int i = t + CE_T * m;
            if (i < Npro)
                st8(lds + ce_nat(i), pro[m]);
        }
        cx.lds_barrier();
    }

#if defined(CE_PROF) && defined(__HIP_DEVICE_COMPILE__)
    unsigned long long prof_acc[8] = {}, prof_t = __builtin_amdgcn_s_memrealtime();
#endif
    for (int j = 0; j < ntile; j++) {
        const int IMG = j & 1;
        cf* const A = lds + IMG * CE_IMG;
        cf* const B = lds + (1 - IMG) * CE_IMG;
        const unsigned imgB = lds0 + (unsigned)((1 - IMG) * CE_IMG * 8);
        const int k0 = (tile0 + j) * L;
        if (!(CE_DBG & 32))
            cx.wait_dma(); // this wave's share of the window (issued a tile ago) has landed
#if (CE_DBG & 128) && defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::"v"(dbg_land[0]), "v"(dbg_land[1]), "v"(dbg_land[2]), "v"(dbg_land[3]));
#endif
        CE_TICK(0);
        if (!(CE_DBG & 4))
            cx.lds_barrier(); // ... and everybody's; everybody has left the other image
        CE_TICK(1);
        cf x[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; n1++)
            x[n1] = ld8(A + n1 * CE_ROW + ce.c_nat);
        cx.wave_sync(); // (lane model: the wave's reads are done before its in-place writes)
        // A2: out[k0 + i] = stream[k0 + i - N] = w[i], i < L   (lib/corr_est_cc_impl.cc:184)
        if (CE_DBG & 1) {
        } else if (k0 + L <= n) {
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) {
                const int lo = CE_T * n1;
                if (lo + CE_T <= L)
                    cx.buf_store64(bout, (unsigned)t * 8u, (unsigned)(k0 + lo) * 8u, x[n1]);
                else if (lo < L) {
                    if (t < L - lo)
                        cx.buf_store64(bout, (unsigned)t * 8u, (unsigned)(k0 + lo) * 8u, x[n1]);
                }
            }
        } else { // last tile of the call: the hardware drops what lies beyond n
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) {
                const int i = t + CE_T * n1;
                if (i < L)
                    cx.buf_store64(bout, (unsigned)(k0 + i) * 8u, 0u, x[n1]);
            }
        }
        // the overlap -- items [L, F) of this window are items [0, N) of the next one -- goes across
        // from the registers that hold it; the new items of the next window by DMA
        if (j + 1 < ntile) {
            constexpr int NOV = NC ? (NC + CE_T - 1) / CE_T + 1 : 5; // slices that can hold items >= L
#pragma unroll
            for (int m = 0; m < NOV; m++) {
                const int n1 = 8 - NOV + m;
                if (CE_T * n1 + CE_T - 1 >= L) {
                    const int d = t + CE_T * n1 - L;
                    if (CE_T * n1 >= L || d >= 0)
                        st8(B + ce_nat(d), x[n1]);
                }
            }
            issue_window(imgB, k0 + L);
        }
        if (p.corr_hist_zero && k0 < N) { // (first tile of a call only)
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++)
                if (k0 - N + t + CE_T * n1 < 0)
                    x[n1] = mk(0.f, 0.f);
        }
        CE_TICK(2);
        if (!(CE_DBG & 64))
            ce.fwd1(A, x);
        CE_TICK(3);
        if (!(CE_DBG & 4))
            cx.lds_barrier();
        CE_TICK(4);
        if (!(CE_DBG & 16)) {
            ce.fwd234(A, x);
#pragma unroll
            for (int k4 = 0; k4 < 8; k4++)
                x[k4] = cx.cmul(x[k4], H[k4]);
            ce.inv432(A, x);
        }
        CE_TICK(5);
        if (!(CE_DBG & 4))
            cx.lds_barrier();
        CE_TICK(6);
        if (!(CE_DBG & 64))
            ce.inv1(A, x);
        // y[i] = corr[k0 + i - N]; A4 mag^2 (:191) and the threshold test (:197) as k_corr4d.h
        const bool interior = (k0 - N >= 0) && (k0 + L <= n);
        const int kb = k0 + t - N; // output index of value n1: kb + 512 n1
        if (CE_DBG & 8) {
            if (x[0].re == 1.2345f)
                xcorr[kb] = x[1];
        } else if (interior && !p.dense_corr) {
            bool any = false;
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) {
                if (CE_T * n1 + CE_T - 1 < N)
                    continue; // never an output
                const float mg = mag2(x[n1]);
                const bool above = !(mg <= p.thresh);
                if (CE_T * n1 >= N)
                    any |= above;
                else
                    any |= above && (t + CE_T * n1 >= N);
            }
            if (cx.ballot(any) != 0ull) {
                unsigned hit = 0;
#pragma unroll
                for (int n1 = 0; n1 < 8; n1++)
                    hit |= (!(mag2(x[n1]) <= p.thresh)) ? (1u << n1) : 0u;
                hit &= vmask_int;
                corr_emit_hits_n<Ctx, 8>(cx, p, hit, vmask_int, x, xcorr, abits, kb, CE_T);
            }
        } else {
            unsigned vmask = 0;
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) {
                const int m = t + CE_T * n1 - N;
                if (m >= 0 && m < L && k0 + m < n)
                    vmask |= 1u << n1;
            }
            if (p.dense_corr) {
#pragma unroll
                for (int n1 = 0; n1 < 8; n1++)
                    if ((vmask >> n1) & 1u)
                        xcorr[kb + CE_T * n1] = x[n1];
            }
            unsigned hit = 0;
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++)
                hit |= (!(mag2(x[n1]) <= p.thresh)) ? (1u << n1) : 0u;
            hit &= vmask;
            if (cx.ballot(hit != 0u) != 0ull)
                corr_emit_hits_n<Ctx, 8>(cx, p, hit, vmask, x, xcorr, abits, kb, CE_T);
        }
        CE_TICK(7);
    }
#if defined(CE_PROF) && defined(__HIP_DEVICE_COMPILE__)
    if (lane == 0) {
        for (int i = 0; i < 8; i++)
            atomicAdd(&g_ce_prof[i], prof_acc[i]);
        atomicAdd(&g_ce_prof[8], (unsigned long long)ntile);
    }
#endif
    // carry the last N stream samples to the next call (set_history(N+1), :95)
    if (seg == p.nseg - 1) {
        cf* ho = p.hist_out + (long)c * N;
        for (int jj = t; jj < N; jj += CE_T) {
            const int s = n - N + jj;
            ho[jj] = (s >= 0) ? xin[s] : hist[N + s];
        }
    }
}

} // namespace aisx
