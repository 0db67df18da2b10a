// k_corr4d.h -- the F = 4096 correlator (k_corr4k.h: same transform, same contract) with the
// window of the NEXT tile fetched by LDS-DMA while the current one is transformed.
//
// What bounded k_corr4_main (profiles/r01_corr_main_pmc.json): every tile started with sixteen
// global loads per thread that nothing could be overlapped with at three waves per SIMD, the
// N-sample overlap of neighbouring windows was read twice (1.38 x the algorithmic read traffic:
// the per-XCD L2 does not keep it), H and the twiddles were re-read from L2 per tile, and
// __syncthreads() waited for the pass-through stores before every first barrier.
//
// Here a workgroup (256 threads, one transform) owns TWO LDS images of the window and walks the
// tiles of its channel segment in order:
//   * image X[j & 1] holds the window of tile j in natural order (row n1 = 256 items, pitch
//     CF4_ROW); the first pass reads it, and writes its result back IN PLACE (every thread's
//     sixteen outputs land in the same sixteen 16-element groups its inputs came from, each
//     group private to a wave);
//   * while tile j is transformed, X[(j+1) & 1] fills up: the last N items of window j ARE the
//     first N of window j + 1 and are copied across from the registers that hold them (the
//     overlap never comes from memory again), the L new items arrive by `buffer_load ... lds`
//     (1 KiB per wave instruction, no VGPRs, out-of-range items read as zero), issued a whole
//     tile ahead of their use;
//   * barriers order LDS traffic only (s_waitcnt lgkmcnt + s_barrier): DMA and the pass-through
//     stores stay in flight across them; one vmcnt(0) per tile, where the next window is needed;
//   * with two workgroups per CU the register budget is 256: H (the thread's 16 spectrum
//     positions) and both twiddle sets stay in VGPRs for the whole kernel; the tile loop is
//     unrolled by two so that the image in hand is a compile-time constant and every LDS address
//     a per-thread index plus an immediate.
// HBM traffic per tile: L items read + L items written = the algorithmic 16 B per sample.
#pragma once
#include <type_traits>
#include "k_corr4k.h"

namespace aisx {

constexpr int CD_IMG = 16 * CF4_ROW;               // complex slots per window image
// build switches (measured variants, MI355X, 4096 x 65536, N = 896, kernel alone; the defaults
// are what the product ships):
//   CD_FOUR_BARRIERS  1 = four barriers per tile (overlap copy and DMA issue held back behind the
//                     first pass's barrier), 0 = five (one at the tile's top).  Four: 1.31-1.33 ms
//                     against 1.27-1.29: the window then has three passes to land instead of a
//                     whole tile, and eight more VGPRs are live across the first pass.
//   CD_UNROLL2        1 = tile loop unrolled by two, the image in hand a compile-time constant (LDS
//                     addresses = per-thread index + immediate, ~80 VALU fewer per tile), 0 = one
//                     loop body.  Unrolled: 1.35 ms against 1.27-1.29 (twice the code).
//   CD_W2_REGS        1 = second / third pass twiddles in registers: 64 VGPR spills, not measured.
#ifndef CD_FOUR_BARRIERS
#define CD_FOUR_BARRIERS 0
#endif
#ifndef CD_UNROLL2
#define CD_UNROLL2 0
#endif
// second / third pass twiddles W_256^{k2 n3}: 1 = in registers (30 VGPRs), 0 = a 2 KB table in LDS
#ifndef CD_W2_REGS
#define CD_W2_REGS 0
#endif
//   CD_WAVE_EXCHANGE  1 = the second LDS exchange of each direction is a wave-level one.  With
//                     thread t = (k1, k2 | n3) = (t >> 4, t & 15) the forward pass over n2 writes
//                     row k1, columns k2 * 16 + n3 for all k2 from thread (k1, n3), and the pass over
//                     n3 reads row k1, columns k2 * 16 + n3 for all n3 from thread (k1, k2): writer and
//                     reader share k1 = t >> 4, i.e. they sit in the same sixteen-lane row of one
//                     wave (the inverse direction mirrors it).  Only the exchanges between the pass
//                     over n1 and the pass over n2 cross waves: three workgroup barriers per tile
//                     instead of five.  0 = a workgroup barrier at all four exchanges (round 2).
//   CD_GENERIC_W_REGS 1 = the run-time-length build (NC = 0) keeps the first / last pass twiddles in registers like the
//                     folded builds: 9 VGPRs spilled to scratch (40 B per lane) + 222 SGPR spills (round 4's build).
//                     0 = it reads them where it uses them: no scratch (round 5; A/B in DESIGN.md section 8).
#ifndef CD_GENERIC_W_REGS
#define CD_GENERIC_W_REGS 0
#endif
#ifndef CD_WAVE_EXCHANGE
#define CD_WAVE_EXCHANGE 1
#endif
constexpr int CD_LDS_ELEMS = 2 * CD_IMG + (CD_W2_REGS ? 0 : 256); // two images (+ the W_256 table)
constexpr int CD_LDS_BYTES = CD_LDS_ELEMS * 8;     // 69632 / 71680 B: two workgroups per CU
constexpr int CD_PIECE = 128;                      // items per DMA wave-instruction (64 lanes x 16 B)

// natural-order slot of window item i (what the DMA writes and the first pass reads)
AISX_HD int cd_nat(int i) { return (i >> 8) * CF4_ROW + (i & 255); }

// NC: the template length as a compile-time constant (every slice predicate of the tile loop
// then folds: which of a thread's sixteen items are pass-through outputs, overlap, correlation
// outputs), or 0 for the run-time version that serves any length.
//
// Barriers per tile: five (CD_FOUR_BARRIERS = 0).  With four, the loop carries no barrier between
// the last inverse pass of tile j (reads image A) and the first forward pass of tile j + 1 (reads
// and rewrites image B in place):
//   * a wave waits for its share of the next window (vmcnt) BEFORE the barrier that ends the
//     fourth pass, so behind that barrier the whole window is in LDS;
//   * what tile j + 1 writes into image A -- the overlap and the DMA of window j + 2 -- is held
//     back until the barrier that ends ITS first pass: by then every wave has left tile j.
template <class Ctx, int NC>
AISX_DI void corr4d_main_body(Ctx& cx, const CorrParams& p)
{
    const int t = cx.tid();
    const int wave = cx.wave_id();
    const int lane = t & 63;
    const int c = cx.by();
    const int seg = cx.bx();
    cf* lds = (cf*)cx.lds();

    const int N = NC ? NC : p.N, L = CF4_F - N, n = p.n;
    const cf* xin = p.in + (long)c * p.in_stride;
    cf* xout = p.out + (long)c * p.out_stride;
    cf* xcorr = p.corr + (long)c * p.corr_stride;
    const cf* hist = p.hist_in + (long)c * N;
    unsigned long long* abits = p.abits + (long)c * p.abits_stride;

    // per-thread constants of the transform, in registers for the whole kernel: first / last pass
    // twiddles W_4096^{k t}, second / third pass twiddles W_256^{k2 n3}, the thread's 16 positions of H
    // (the run-time-length build, NC = 0, has sixteen slice predicates live as lane masks on top of this and
    // does not fit 256 VGPRs with the first / last pass twiddles resident: it reads them from the 32 KB
    // table -- L1 / L2 hits -- where it uses them; the folded builds keep them)
    constexpr bool W_REGS = NC != 0 || CD_GENERIC_W_REGS;
    cf w[16], H[16];
    w[0] = mk(1.f, 0.f);
#pragma unroll
    for (int k = 1; k < 16; k++)
        w[k] = W_REGS ? p.wtab[(k * t) & (CF4_F - 1)] : mk(0.f, 0.f);
    auto tw1 = [&](int k) -> cf { return W_REGS ? w[k] : p.wtab[(k * t) & (CF4_F - 1)]; };
#if CD_W2_REGS
    cf w2[16];
    w2[0] = mk(1.f, 0.f);
#pragma unroll
    for (int k = 1; k < 16; k++)
        w2[k] = p.wtab[(16 * k * (t & 15)) & (CF4_F - 1)];
    auto tw2 = [&](int k2) -> cf { return w2[k2]; };
#else
    cf* const ldsT = lds + 2 * CD_IMG;
    ldsT[t] = p.wtab[(16 * (t >> 4) * (t & 15)) & (CF4_F - 1)]; // W_256^{k2 n3}, index k2*16+n3
    const cf* const myT = ldsT + (t & 15);
    auto tw2 = [&](int k2) -> cf { return ld8(myT + k2 * 16); };
#endif
#pragma unroll
    for (int k3 = 0; k3 < 16; k3++)
        H[k3] = p.Hpos[t * 16 + k3];

    unsigned vmask_int = 0; // value n1 of this thread is window item t + 256 n1: an output iff >= N
#pragma unroll
    for (int n1 = 0; n1 < 16; n1++)
        if (t + CF4_T * n1 >= N)
            vmask_int |= 1u << n1;

    const auto bin = cx.make_buf(xin, (unsigned)n * 8u);
    const auto bout = cx.make_buf(xout, (unsigned)n * 8u);
    const int P0 = N / CD_PIECE; // first DMA piece that holds new items (it may hold a few old ones too)
    const unsigned lds0 = cx.lds_addr(lds);

    // the new items of the window of the tile whose outputs start at k0: pieces P0 .. 31, spread
    // over the four waves; item i of the window is stream item k0 - N + i
    auto issue_window = [&](unsigned img, int k0) {
        for (int pc = P0 + wave; pc < CF4_F / CD_PIECE; pc += 4) {
            const unsigned dst = img + (unsigned)((pc >> 1) * (CF4_ROW * 8) + (pc & 1) * (CD_PIECE * 8));
            // (a negative stream index -- first tile of a call, piece P0 -- wraps far out of range: zeros)
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

    // ---- first window of the segment: new items by DMA, the N before them by plain loads
    // (history of the block where the stream index is negative, lib/corr_est_cc_impl.cc:180-188)
    if (ntile > 0) {
        const int k0 = tile0 * L;
        issue_window(lds0, k0);
        // (an odd N splits a 16-byte DMA pair between stream items -1 and 0 in the first window of
        // a call; what the hardware returns for the half that wraps is not relied upon: item N
        // is then written here as well)
        const int Npro = N + (N & 1);
        cf pro[8];
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int i = t + CF4_T * m, s = k0 - N + i;
            pro[m] = mk(0.f, 0.f);
            if (i < Npro)
                pro[m] = (s < 0) ? hist[N + s] : ((s < n) ? xin[s] : mk(0.f, 0.f));
        }
        cx.wait_dma();
        cx.lds_barrier(); // every wave's pieces have landed: piece P0 may overlap the items below N
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int i = t + CF4_T * m;
            if (i < Npro)
                st8(lds + cd_nat(i), pro[m]);
        }
        cx.lds_barrier();
    }

    // one tile in image IMG (a compile-time constant: every LDS address is a per-thread index
    // plus an immediate), the other image filling up for the next one
    auto tile = [&](auto IMGC, int j) {
#if CD_UNROLL2
        constexpr int IMG = decltype(IMGC)::value;
#else
        const int IMG = j & 1;
#endif
        cf* const A = lds + IMG * CD_IMG;
        cf* const B = lds + (1 - IMG) * CD_IMG;
        const unsigned imgB = lds0 + (unsigned)((1 - IMG) * CD_IMG * 8);
        const int k0 = (tile0 + j) * L;
#if !CD_FOUR_BARRIERS
        cx.wait_dma();    // this wave's share of the window (issued a tile ago) has landed
        cx.lds_barrier(); // ... and everybody's; everybody has left the other image
#endif
        cf x[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; n1++)
            x[n1] = ld8(A + n1 * CF4_ROW + t);
        cx.wave_sync(); // (lane model: the wave's reads are done before its in-place writes)
        // A2: out[k0 + i] = stream[k0 + i - N] = w[i], i < L   (lib/corr_est_cc_impl.cc:184)
        if (k0 + L <= n) {
            // (whole slices of 256 items: no predicate; the slice L ends in: its first L % 256 lanes)
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                const int lo = CF4_T * n1;
                if (lo + CF4_T <= L)
                    cx.buf_store64(bout, (unsigned)t * 8u, (unsigned)(k0 + lo) * 8u, x[n1]);
                else if (lo < L) {
                    if (t < L - lo)
                        cx.buf_store64(bout, (unsigned)t * 8u, (unsigned)(k0 + lo) * 8u, x[n1]);
                }
            }
        } else { // last tile of the call: the hardware drops what lies beyond n
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                const int i = t + CF4_T * n1;
                if (i < L)
                    cx.buf_store64(bout, (unsigned)(k0 + i) * 8u, 0u, x[n1]);
            }
        }
        // the overlap -- items [L, F) of this window are items [0, N) of the next one -- waits in
        // registers until the other image is free (after the barrier below)
        constexpr int NOV = NC ? (NC + CF4_T - 1) / CF4_T + 1 : 9; // slices that can hold items >= L
        auto next_window = [&](const cf* src) { // src[m] = this window's item t + 256 (16 - NOV + m)
#pragma unroll
            for (int m = 0; m < NOV; m++) {
                const int n1 = 16 - NOV + m;
                // (slices wholly below L do nothing; items at or above piece P0 arrive by DMA as
                // well, with the same value: copying them too saves the test)
                if (CF4_T * n1 + CF4_T - 1 >= L) {
                    const int d = t + CF4_T * n1 - L;
                    if (CF4_T * n1 >= L || d >= 0)
                        st8(B + cd_nat(d), src[m]);
                }
            }
            issue_window(imgB, k0 + L);
        };
#if CD_FOUR_BARRIERS
        cf ov[NOV];
#pragma unroll
        for (int m = 0; m < NOV; m++)
            ov[m] = x[16 - NOV + m];
#else
        if (j + 1 < ntile)
            next_window(&x[16 - NOV]);
#endif
        if (p.corr_hist_zero && k0 < N) { // (first tile of a call only)
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++)
                if (k0 - N + t + CF4_T * n1 < 0)
                    x[n1] = mk(0.f, 0.f);
        }
        // ---- forward: three radix-16 passes (k_corr4k.h: cf4_forward) with the twiddles in registers
        dft16<false>(cx, x);
#pragma unroll
        for (int k1 = 1; k1 < 16; k1++)
            x[k1] = cmul_fma(x[k1], tw1(k1));
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++)
            st8(A + cf4_pos(k1, t), x[k1]);
        cx.lds_barrier();
#if CD_FOUR_BARRIERS
        if (j + 1 < ntile) // every wave has left the previous tile: its image takes the next window
            next_window(ov);
#endif
        {
            const int k1 = t >> 4, n3 = t & 15;
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++)
                x[n2] = ld8(A + cf4_pos(k1, n2 * 16 + n3));
            dft16<false>(cx, x);
#pragma unroll
            for (int k2 = 1; k2 < 16; k2++)
                x[k2] = cmul_fma(x[k2], tw2(k2));
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++)
                st8(A + cf4_pos(k1, k2 * 16 + n3), x[k2]);
        }
#if CD_WAVE_EXCHANGE
        cx.wave_lds_sync(); // writers (k1, n3) and readers (k1, k2) share k1: one wave
#else
        cx.lds_barrier();
#endif
        {
            const int k1 = t >> 4, k2 = t & 15, swz = k2 >> 1;
            const int base = k1 * CF4_ROW + k2 * 16;
#pragma unroll
            for (int pr = 0; pr < 8; pr++) {
                const int ch = base + 2 * (pr ^ swz);
                ld16(A + ch, x[2 * pr], x[2 * pr + 1]);
            }
            dft16<false>(cx, x);
            // spectrum x H, inverse radix-16
#pragma unroll
            for (int k3 = 0; k3 < 16; k3++)
                x[k3] = cmul_fma(x[k3], H[k3]);
            dft16<true>(cx, x);
#pragma unroll
            for (int pr = 0; pr < 8; pr++) {
                const int ch = base + 2 * (pr ^ swz);
                st16(A + ch, x[2 * pr], x[2 * pr + 1]);
            }
        }
#if CD_WAVE_EXCHANGE
        cx.wave_lds_sync(); // (the mirror image: written by (k1, k2), read by (k1, n3))
#else
        cx.lds_barrier();
#endif
        {
            const int k1 = t >> 4, n3 = t & 15;
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) {
                cf a = ld8(A + cf4_pos(k1, k2 * 16 + n3));
                x[k2] = (k2 == 0) ? a : cmul_conj_fma(a, tw2(k2));
            }
            dft16<true>(cx, x);
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++)
                st8(A + cf4_pos(k1, n2 * 16 + n3), x[n2]);
        }
#if CD_FOUR_BARRIERS
        cx.wait_dma();    // this wave's share of the next window (issued three passes ago) has landed
#endif
        cx.lds_barrier(); // ... and, behind this barrier, everybody's
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++) {
            cf a = ld8(A + cf4_pos(k1, t));
            x[k1] = (k1 == 0) ? a : cmul_conj_fma(a, tw1(k1));
        }
        dft16<true>(cx, x);
        // y[i] = corr[k0 + i - N]; A4 mag^2 (:191) and the threshold test (:197).  Value n1 of a
        // thread is window item t + 256 n1: a correlation output iff >= N (and, in the first and
        // last tile of a call, inside the call's items).  Interior tiles only ask "does any lane
        // have a hit": the per-value tests stay lane masks in scalar registers; a wave with a hit
        // (rare) then builds its 16-bit masks and walks them.
        const bool interior = (k0 - N >= 0) && (k0 + L <= n);
        const int kb = k0 + t - N; // output index of value n1: kb + 256 n1
        if (interior && !p.dense_corr) {
            bool any = false;
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                if (CF4_T * n1 + CF4_T - 1 < N)
                    continue; // never an output
                const float mg = mag2(x[n1]);
                const bool above = !(mg <= p.thresh);
                if (CF4_T * n1 >= N)
                    any |= above;
                else
                    any |= above && (t + CF4_T * n1 >= N);
            }
            if (cx.ballot(any) != 0ull) {
                unsigned hit = 0;
#pragma unroll
                for (int n1 = 0; n1 < 16; n1++)
                    hit |= (!(mag2(x[n1]) <= p.thresh)) ? (1u << n1) : 0u;
                hit &= vmask_int;
                corr_emit_hits(cx, p, hit, vmask_int, x, xcorr, abits, kb, CF4_T);
            }
        } else {
            unsigned vmask = 0;
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                const int m = t + CF4_T * n1 - N;
                if (m >= 0 && m < L && k0 + m < n)
                    vmask |= 1u << n1;
            }
            if (p.dense_corr) {
#pragma unroll
                for (int n1 = 0; n1 < 16; n1++)
                    if ((vmask >> n1) & 1u)
                        xcorr[kb + CF4_T * n1] = x[n1];
            }
            unsigned hit = 0;
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++)
                hit |= (!(mag2(x[n1]) <= p.thresh)) ? (1u << n1) : 0u;
            hit &= vmask;
            if (cx.ballot(hit != 0u) != 0ull)
                corr_emit_hits(cx, p, hit, vmask, x, xcorr, abits, kb, CF4_T);
        }
    };

#if CD_UNROLL2
    for (int j = 0; j < ntile; j += 2) {
        tile(std::integral_constant<int, 0>{}, j);
        if (j + 1 < ntile)
            tile(std::integral_constant<int, 1>{}, j + 1);
    }
#else
    for (int j = 0; j < ntile; j++)
        tile(0, j);
#endif
    // carry the last N stream samples to the next call (set_history(N+1), :95)
    if (seg == p.nseg - 1) {
        cf* ho = p.hist_out + (long)c * N;
        for (int jj = t; jj < N; jj += CF4_T) {
            const int s = n - N + jj;
            ho[jj] = (s >= 0) ? xin[s] : hist[N + s];
        }
    }
}

} // namespace aisx
