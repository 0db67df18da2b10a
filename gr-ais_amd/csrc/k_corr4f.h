// k_corr4f.h -- the F = 4096 correlator on 512 threads x 8 points (k_corr4e.h: same transform, same
// contract) with the window of the next tile fetched into REGISTERS instead of a second LDS image.
//
// What bounded k_corr4e_main (round 6, CE_DBG builds that leave parts of the tile loop out, 4096 x
// 65536, N = 896, steady clocks): passes alone 0.64 ms per launch and the LDS 85 % busy in them --
// six exchanges of 32 KB, whose stores run at 85 B/clk/CU against 256 B/clk for the loads; the
// pass-through stores beside them +0.05 ms; the window by LDS-DMA beside them +0.18 ms, and not
// through latency (no wave ever waits for its pieces) but through the LDS: the same bytes loaded
// into registers nobody reads cost 0.08-0.11 ms less.  A `buffer_load ... lds` piece of 1 KiB
// holds the LDS ~30 cycles, more than twice a ds_write_b128 of the same bytes.
//
// Here the window never passes through the LDS on its way in:
//   * value n1 of thread t is window item i = t + 512 n1.  The NEW items (i >= N: stream item
//     k0 - N + i) of tile j + 1 are loaded straight into seven register pairs while tile j is
//     transformed (raw buffer loads, 512 contiguous bytes per wave instruction, out-of-range
//     items read as zero); the compiler's own vmcnt bookkeeping waits where they are first used;
//   * the overlap (items [L, F) of window j = items [0, N) of window j + 1) goes through a small
//     LDS buffer of N items, double-buffered: written from registers at the top of tile j, read
//     back by the threads that own those items at the end of tile j, two barriers later;
//   * one window image: the first pass writes columns only its own wave reads back in the last
//     pass, so the tile loop needs no barrier at its top -- two barriers per tile, not three;
//   * 36 864 B (image) + 4 032 B (W_512 / W_64 tables) + 2 x 8 N B (overlap): 55 KB at N = 896.
// The second / third pass twiddles come from the LDS tables (loads are the cheap half of the
// LDS); the registers they held take the prefetched items.
#pragma once
#include "k_corr4e.h"

namespace aisx {

// overlap buffer: slots per buffer for a template of N items (even: 16-byte aligned buffers)
AISX_HD int cfz_ov_slots(int N) { return (N + 1) & ~1; }
AISX_HD int cfz_lds_bytes(int N) { return (CE_IMG + 7 * 64 + 7 * 8 + 2 * cfz_ov_slots(N)) * 8; }

// second pass twiddles W_512^{k2 r2}: 1 = in registers (the folded builds have the room: 120 VGPRs), 0 = the LDS table
#ifndef CFZ_W2_REGS
#define CFZ_W2_REGS 1
#endif

template <class Ctx, int NC>
AISX_DI void corr4f_main_body(Ctx& cx, const CorrParams& p)
{
    cf* lds = (cf*)cx.lds();
    cf* const A = lds;
    Ce<Ctx, (CFZ_W2_REGS != 0) && (NC != 0), false> ce(cx, p.wtab, lds + CE_IMG);
    const int t = ce.t;
    const int c = cx.by();
    const int seg = cx.bx();

    const int N = NC ? NC : p.N, L = CF4_F - N, n = p.n;
    cf* const ov0 = lds + CE_IMG + 7 * 64 + 7 * 8;
    const int ovs = cfz_ov_slots(N);
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

    const int tile0 = seg * p.tiles_per_seg;
    int ntile = p.tiles_per_seg;
    {
        const int left = (n - tile0 * L + L - 1) / L;
        ntile = ntile < left ? ntile : left;
    }

    // ---- first window of the segment: plain loads (history of the block where the stream index is
    // negative, lib/corr_est_cc_impl.cc:180-188)
    cf x[8];
    if (ntile > 0) {
        const int k0 = tile0 * L;
#pragma unroll
        for (int n1 = 0; n1 < 8; n1++) {
            const int s = k0 - N + t + CE_T * n1;
            x[n1] = (s < 0) ? hist[N + s] : ((s < n) ? xin[s] : mk(0.f, 0.f));
        }
        // (the loads are waited for HERE: the compiler's count of what is in flight is then empty at the head of the
        // tile loop, and the only wait inside the loop is the one for the prefetched items at its end)
#pragma unroll
        for (int n1 = 0; n1 < 8; n1++) {
            cx.pin(x[n1].re);
            cx.pin(x[n1].im);
        }
    }
    // (the same for the thread's constants: their first use is inside the loop)
#pragma unroll
    for (int k = 0; k < 8; k++) {
        cx.pin(H[k].re);
        cx.pin(H[k].im);
        cx.pin(ce.w1[k].re);
        cx.pin(ce.w1[k].im);
        cx.pin(ce.w2[k].re);
        cx.pin(ce.w2[k].im);
    }
    cx.lds_barrier(); // (the tables of Ce)

    for (int j = 0; j < ntile; j++) {
        const int k0 = (tile0 + j) * L;
        const bool more = j + 1 < ntile;
        cf* const ovw = ov0 + ((j + 1) & 1) * ovs; // overlap for the next tile
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
        // the new items of the next window: stream items k0 + L - N + i, i >= N.  (Behind the stores: they read x, whose
        // first fill the compiler still counts as in flight at the loop's head -- a wait there, ahead of these loads, is free;
        // behind them it would wait for them.)
        cf nx[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; n1++) {
            nx[n1] = mk(0.f, 0.f);
            if (CE_T * n1 + CE_T - 1 >= N && more && !(CE_DBG & 2) && !(CE_DBG & 512)) {
                const int i = t + CE_T * n1;
                if (CE_T * n1 >= N || i >= N)
                    nx[n1] = cx.buf_load64(bin, (unsigned)(k0 + L - N + i) * 8u, 0u);
            }
        }
#if (CE_DBG & 512) && defined(__HIP_DEVICE_COMPILE__)
        // (timing only: the same bytes as 16-byte loads, two slices per instruction)
        if (more) {
#pragma unroll
            for (int n1 = 0; n1 < 8; n1 += 2) {
                if (CE_T * n1 + 2 * CE_T - 1 >= N) {
                    typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
                    const v4u_ d = __builtin_amdgcn_raw_buffer_load_b128(
                        __builtin_amdgcn_make_buffer_rsrc((void*)bin.base, 0, (int)bin.nbytes, 0x00020000),
                        (int)((unsigned)(k0 + L - N + CE_T * n1 + 2 * t) * 8u), 0, AISX_STORE_AUX);
                    nx[n1] = mk(__uint_as_float(d.x), __uint_as_float(d.y));
                    nx[n1 + 1] = mk(__uint_as_float(d.z), __uint_as_float(d.w));
                }
            }
        }
#endif
        // the overlap: items [L, F) of this window are items [0, N) of the next one
        if (more) {
            constexpr int NOV = NC ? (NC + CE_T - 1) / CE_T + 1 : 5; // slices that can hold items >= L
#pragma unroll
            for (int m = 0; m < NOV; m++) {
                const int n1 = 8 - NOV + m;
                if (CE_T * n1 + CE_T - 1 >= L) {
                    const int d = t + CE_T * n1 - L;
                    if (CE_T * n1 >= L || d >= 0)
                        st8(ovw + d, x[n1]);
                }
            }
        }
        if (p.corr_hist_zero && k0 < N) { // (first tile of a call only)
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++)
                if (k0 - N + t + CE_T * n1 < 0)
                    x[n1] = mk(0.f, 0.f);
        }
        if (!(CE_DBG & 64))
            ce.fwd1(A, x);
        cx.lds_barrier();
        if (!(CE_DBG & 16)) {
            ce.fwd234(A, x);
#pragma unroll
            for (int k4 = 0; k4 < 8; k4++)
                x[k4] = cx.cmul(x[k4], H[k4]);
            ce.inv432(A, x);
        }
        cx.lds_barrier();
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
        // the next window: its first N items from the overlap buffer (written at the top of this
        // tile, two barriers ago), the rest from the registers they were loaded into
        if (more && !(CE_DBG & 256)) {
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) {
                const int i = t + CE_T * n1;
                if (CE_T * n1 >= N)
                    x[n1] = nx[n1];
                else if (CE_T * n1 + CE_T - 1 < N)
                    x[n1] = ld8(ovw + i);
                else
                    x[n1] = (i < N) ? ld8(ovw + (i < N ? i : 0)) : nx[n1];
            }
        }
    }
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
