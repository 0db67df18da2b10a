// k_corr2d.h -- the F = 2048 correlator (k_corr.h: same transform, same contract, templates of up
// to 512 samples) built the way k_corr4d.h builds the F = 4096 one: a workgroup (128 threads, one
// transform) owns TWO LDS images of the window and walks the tiles of its channel segment;
//   * while tile j is transformed in place in one image, the window of tile j + 1 fills the other:
//     the items below its first DMA piece are copied across from the registers that hold them, the
//     rest arrives by `buffer_load_dwordx4 ... lds` (one 128-item row per wave instruction, no
//     VGPRs, out-of-range items read as zero), issued a whole tile ahead of its use;
//   * barriers order LDS traffic only; DMA and pass-through stores stay in flight across them;
//   * H (the thread's 16 spectrum positions) and the first-pass twiddles W_2048^{k1 t} live in VGPRs
//     for the whole kernel (k_corr_main re-reads both from L2 in every tile: that latency, at three
//     waves per SIMD, is what bounded it);
//   * the radix-8 pass takes its two 8-point transforms per thread from the rows its OWN wave wrote in
//     the pass before (q = 128 wave + 64 h + lane instead of t + 128 h): the second LDS exchange of
//     each direction is then wave-private, as in k_corr4d.h -- three workgroup barriers per tile.
// LDS: 2 x 17 408 B + 1 KB of W_128 twiddles = 35 840 B, four workgroups (eight waves) per CU.
// HBM traffic per tile: L + (N mod 128 ... 128) items read, L written.
#pragma once
#include "k_corr.h"

namespace aisx {

constexpr int C2_IMG = 16 * CF_ROW;                // complex slots per window image
constexpr int C2_LDS_ELEMS = 2 * C2_IMG + 128;     // two images + the W_128^{k2 n3} table
constexpr int C2_LDS_BYTES = C2_LDS_ELEMS * 8;     // 35 840
constexpr int C2_PIECE = 128;                      // items per DMA wave-instruction = one image row

// natural-order slot of window item i (what the DMA writes and the first pass reads)
AISX_HD int c2_nat(int i) { return (i >> 7) * CF_ROW + (i & 127); }

// NC: the template length as a compile-time constant, or 0 for the run-time version
template <class Ctx, int NC>
AISX_DI void corr2d_main_body(Ctx& cx, const CorrParams& p)
{
    const int t = cx.tid();
    const int wave = cx.wave_id();
    const int lane = t & 63;
    const int c = cx.by();
    const int seg = cx.bx();
    cf* lds = (cf*)cx.lds();

    const int N = NC ? NC : p.N, L = CF_F - N, n = p.n;
    const cf* xin = p.in + (long)c * p.in_stride;
    cf* xout = p.out + (long)c * p.out_stride;
    cf* xcorr = p.corr + (long)c * p.corr_stride;
    const cf* hist = p.hist_in + (long)c * N;
    unsigned long long* abits = p.abits + (long)c * p.abits_stride;

    // the two 8-point transforms of the radix-8 pass this thread owns: rows its own wave wrote
    const int q0 = CF_T * wave + lane, q1 = q0 + 64; // q = k1 * 16 + k2
    // per-thread constants, in registers for the whole kernel
    cf w[16], H[16];
    w[0] = mk(1.f, 0.f);
#pragma unroll
    for (int k = 1; k < 16; k++)
        w[k] = p.wtab[(k * t) & (CF_F - 1)];
#pragma unroll
    for (int k3 = 0; k3 < 8; k3++) {
        H[k3] = p.Hpos[q0 * 8 + k3];
        H[8 + k3] = p.Hpos[q1 * 8 + k3];
    }
    cf* const ldsT = lds + 2 * C2_IMG;
    ldsT[t] = p.wtab[(16 * (t >> 3) * (t & 7)) & (CF_F - 1)]; // W_128^{k2 n3}, index k2*8+n3
    const cf* const myT = ldsT + (t & 7);
    auto tw2 = [&](int k2) -> cf { return ld8(myT + k2 * 8); };

    // value n1 of this thread is window item t + 128 n1: an output iff >= N.  Built where a hit needs it (from
    // a copy of t the optimiser cannot trace: kept across the tile loop it cost the N = 112 build a spilled VGPR)
    auto vmask_of = [&]() -> unsigned {
        int tt = t;
        cx.pin(tt);
        unsigned m = 0;
#pragma unroll
        for (int n1 = 0; n1 < 16; n1++)
            if (tt + CF_T * n1 >= N)
                m |= 1u << n1;
        return m;
    };

    const auto bin = cx.make_buf(xin, (unsigned)n * 8u);
    const auto bout = cx.make_buf(xout, (unsigned)n * 8u);
    const int P0 = N / C2_PIECE; // first DMA piece that holds new items (it may hold old ones too)
    const unsigned lds0 = cx.lds_addr(lds);

    // the window of the tile whose outputs start at k0, from piece P0 on, spread over the two
    // waves; item i of the window is stream item k0 - N + i
    auto issue_window = [&](unsigned img, int k0) {
        for (int pc = P0 + wave; pc < CF_F / C2_PIECE; pc += 2) {
            const unsigned dst = img + (unsigned)(pc * (CF_ROW * 8));
            // (a negative stream index -- first tile of a call -- wraps far out of range: zeros)
            const unsigned off = (unsigned)(k0 - N + C2_PIECE * pc + 2 * lane) * 8u;
            cx.dma16(bin, off, dst);
        }
    };

    const int tile0 = seg * p.tiles_per_seg;
    int ntile = p.tiles_per_seg;
    {
        const int left = (n - tile0 * L + L - 1) / L;
        ntile = ntile < left ? ntile : left;
    }

    // ---- first window of the segment: from piece P0 on by DMA, what lies before stream item
    // k0 (history of the block where the stream index is negative, lib/corr_est_cc_impl.cc:180-188)
    // by plain loads; an odd N splits a 16-byte DMA pair between items -1 and 0: item N is written
    // here as well
    if (ntile > 0) {
        const int k0 = tile0 * L;
        issue_window(lds0, k0);
        const int Npro = N + (N & 1);
        cf pro[5];
#pragma unroll
        for (int m = 0; m < 5; m++) {
            const int i = t + CF_T * m, s = k0 - N + i;
            pro[m] = mk(0.f, 0.f);
            if (i < Npro)
                pro[m] = (s < 0) ? hist[N + s] : ((s < n) ? xin[s] : mk(0.f, 0.f));
        }
        cx.wait_dma();
        cx.lds_barrier(); // every wave's pieces have landed: piece P0 may overlap the items below N
#pragma unroll
        for (int m = 0; m < 5; m++) {
            const int i = t + CF_T * m;
            if (i < Npro)
                st8(lds + c2_nat(i), pro[m]);
        }
        cx.lds_barrier();
    }

    for (int j = 0; j < ntile; j++) {
        const int IMG = j & 1;
        cf* const A = lds + IMG * C2_IMG;
        cf* const B = lds + (1 - IMG) * C2_IMG;
        const unsigned imgB = lds0 + (unsigned)((1 - IMG) * C2_IMG * 8);
        const int k0 = (tile0 + j) * L;
        cx.wait_dma();    // this wave's share of the window (issued a tile ago) has landed
        cx.lds_barrier(); // ... and everybody's; everybody has left the other image
        cf x[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; n1++)
            x[n1] = ld8(A + n1 * CF_ROW + t);
        cx.wave_sync(); // (lane model: the wave's reads are done before its in-place writes)
        // A2: out[k0 + i] = stream[k0 + i - N] = w[i], i < L   (lib/corr_est_cc_impl.cc:184)
        if (k0 + L <= n) {
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                const int lo = CF_T * n1;
                if (lo + CF_T <= L)
                    cx.buf_store64(bout, (unsigned)t * 8u, (unsigned)(k0 + lo) * 8u, x[n1]);
                else if (lo < L) {
                    if (t < L - lo)
                        cx.buf_store64(bout, (unsigned)t * 8u, (unsigned)(k0 + lo) * 8u, x[n1]);
                }
            }
        } else { // last tile of the call: the hardware drops what lies beyond n
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                const int i = t + CF_T * n1;
                if (i < L)
                    cx.buf_store64(bout, (unsigned)(k0 + i) * 8u, 0u, x[n1]);
            }
        }
        // the overlap -- items [L, F) of this window are items [0, N) of the next one
        if (j + 1 < ntile) {
            constexpr int NOV = NC ? (NC + CF_T - 1) / CF_T + 1 : 5; // slices that can hold items >= L
#pragma unroll
            for (int m = 0; m < NOV; m++) {
                const int n1 = 16 - NOV + m;
                if (CF_T * n1 + CF_T - 1 >= L) {
                    // (items from piece P0 on arrive by DMA: for the intended preamble, N = 112, that is all of them)
                    const int d = t + CF_T * n1 - L;
                    if (d >= 0 && d < C2_PIECE * P0)
                        st8(B + c2_nat(d), x[n1]);
                }
            }
            issue_window(imgB, k0 + L);
        }
        if (p.corr_hist_zero && k0 < N) { // (first tile of a call only)
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++)
                if (k0 - N + t + CF_T * n1 < 0)
                    x[n1] = mk(0.f, 0.f);
        }
        // ---- forward: radix-16, radix-16, radix-8 (k_corr.h: cf_forward) with the twiddles in registers
        dft16<false>(cx, x);
#pragma unroll
        for (int k1 = 1; k1 < 16; k1++)
            x[k1] = cmul_fma(x[k1], w[k1]);
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++)
            st8(A + cf_pos(k1, t), x[k1]);
        cx.lds_barrier();
        {
            const int k1 = t >> 3, n3 = t & 7;
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++)
                x[n2] = ld8(A + cf_pos(k1, n2 * 8 + n3));
            dft16<false>(cx, x);
#pragma unroll
            for (int k2 = 1; k2 < 16; k2++)
                x[k2] = cmul_fma(x[k2], tw2(k2));
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++)
                st8(A + cf_pos(k1, k2 * 8 + n3), x[k2]);
        }
        cx.wave_lds_sync(); // rows 8 wave .. 8 wave + 7 were written by this wave and are read by it
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int q = h ? q1 : q0, k1 = q >> 4, k2 = q & 15, swz = (k2 >> 2) & 3;
            const int base = k1 * CF_ROW + k2 * 8;
            cf v[8];
#pragma unroll
            for (int pr = 0; pr < 4; pr++)
                ld16(A + base + 2 * (pr ^ swz), v[2 * pr], v[2 * pr + 1]);
            dft8<false>(cx, v);
            // spectrum x H, inverse radix-8
#pragma unroll
            for (int k3 = 0; k3 < 8; k3++)
                v[k3] = cmul_fma(v[k3], H[8 * h + k3]);
            dft8<true>(cx, v);
#pragma unroll
            for (int pr = 0; pr < 4; pr++)
                st16(A + base + 2 * (pr ^ swz), v[2 * pr], v[2 * pr + 1]);
        }
        cx.wave_lds_sync();
        {
            const int k1 = t >> 3, n3 = t & 7;
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) {
                cf a = ld8(A + cf_pos(k1, k2 * 8 + n3));
                x[k2] = (k2 == 0) ? a : cmul_conj_fma(a, tw2(k2));
            }
            dft16<true>(cx, x);
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++)
                st8(A + cf_pos(k1, n2 * 8 + n3), x[n2]);
        }
        cx.lds_barrier();
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++) {
            cf a = ld8(A + cf_pos(k1, t));
            x[k1] = (k1 == 0) ? a : cmul_conj_fma(a, w[k1]);
        }
        dft16<true>(cx, x);
        // y[i] = corr[k0 + i - N]; A4 mag^2 (:191) and the threshold test (:197), as in k_corr4d.h
        const bool interior = (k0 - N >= 0) && (k0 + L <= n);
        const int kb = k0 + t - N; // output index of value n1: kb + 128 n1
        if (interior && !p.dense_corr) {
            bool any = false;
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                if (CF_T * n1 + CF_T - 1 < N)
                    continue; // never an output
                const float mg = mag2(x[n1]);
                const bool above = !(mg <= p.thresh);
                if (CF_T * n1 >= N)
                    any |= above;
                else
                    any |= above && (t + CF_T * n1 >= N);
            }
            if (cx.ballot(any) != 0ull) {
                unsigned hit = 0;
#pragma unroll
                for (int n1 = 0; n1 < 16; n1++)
                    hit |= (!(mag2(x[n1]) <= p.thresh)) ? (1u << n1) : 0u;
                const unsigned vmask_int = vmask_of();
                hit &= vmask_int;
                corr_emit_hits(cx, p, hit, vmask_int, x, xcorr, abits, kb, CF_T);
            }
        } else {
            unsigned vmask = 0;
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                const int m = t + CF_T * n1 - N;
                if (m >= 0 && m < L && k0 + m < n)
                    vmask |= 1u << n1;
            }
            if (p.dense_corr) {
#pragma unroll
                for (int n1 = 0; n1 < 16; n1++)
                    if ((vmask >> n1) & 1u)
                        xcorr[kb + CF_T * n1] = x[n1];
            }
            unsigned hit = 0;
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++)
                hit |= (!(mag2(x[n1]) <= p.thresh)) ? (1u << n1) : 0u;
            hit &= vmask;
            if (cx.ballot(hit != 0u) != 0ull)
                corr_emit_hits(cx, p, hit, vmask, x, xcorr, abits, kb, CF_T);
        }
    }
    // carry the last N stream samples to the next call (set_history(N+1), :95)
    if (seg == p.nseg - 1) {
        cf* ho = p.hist_out + (long)c * N;
        for (int jj = t; jj < N; jj += CF_T) {
            const int s = n - N + jj;
            ho[jj] = (s >= 0) ? xin[s] : hist[N + s];
        }
    }
}

} // namespace aisx
