// k_freqsync.h -- square_and_fft_sync_cc (python/gmsk_sync.py:14-37) around
// ais.freqest (lib/freqest_impl.cc:57-88).
//
//   fs_est_body : per (channel, 1024-vector): square (gmsk_sync.py:22,30-31),
//                 1024-point forward FFT with fftshift (:23-24) done by ONE wave
//                 (16 points per lane in VGPRs, 16 x 16 x 4, two LDS exchanges),
//                 |X| = hypot in double as glibc does, then freqest's peak-pair
//                 search (freqest_impl.cc:74-83: first strict maximum of
//                 |X[j]| + |X[j+offset]|) with a wave arg-max.  Emits maxpos, or
//                 -1 when no bin pair had positive energy (the reference then
//                 re-uses the previous vector's maxpos, :68 vs :74).
//   fs_mix_body : per 16 channels: resolves the stale-maxpos rule in sequence,
//                 f = (float(maxpos) - fftlen/2) * binsize / 2 (:84), then
//                 repeat -> frequency_modulator_fc -> multiply_cc
//                 (gmsk_sync.py:26-28,33).  The NCO phase is GNU Radio's float
//                 accumulator with its fmod wrap, a strict recurrence, so one
//                 lane per channel walks it and the other waves do the
//                 sin/cos (gr::fxpt's table, as frequency_modulator_fc) + complex
//                 multiply with coalesced traffic.
// Only fftlen = 1024 (the value the reference uses, python/radio.py:60) is
// implemented by fs_est_body.
#pragma once
#include "aisx_common.h"
#include "k_fft.h"

namespace aisx {

constexpr int FS_F = 1024;
// (measured, round 2: two waves per workgroup -- 18 KB of LDS, and 64 VGPRs with a cap -- so that
// the estimates of the next step might slip in beside the AGC pass from a stream of their own:
// they still start only when that pass has been dispatched; -DFS_WAVES_PER_WG=2 rebuilds it)
#ifndef FS_WAVES_PER_WG
#define FS_WAVES_PER_WG 4
#endif
constexpr int FS_WAVES = FS_WAVES_PER_WG; // waves per workgroup, each on its own vectors
// consecutive 1024-vectors a wave takes, one after the other: the first-pass twiddles (fifteen 8-byte gathers per lane,
// as many bytes from L2 as the vector itself) are loaded once for all of them, and the samples of the next vector
// are in flight while this one is transformed
#ifndef FS_VPW
#define FS_VPW 4
#endif
constexpr int FS_VEC_PER_WG = FS_WAVES * FS_VPW;
// 1 = the next vector's samples are requested before this one's transform (32 more VGPRs live), 0 = behind its search
#ifndef FS_PREFETCH
#define FS_PREFETCH 1
#endif
constexpr int FS_T = 64 * FS_WAVES;
constexpr int FS_ROW = 68;            // LDS row pitch (complex) per k1 row
constexpr int FS_WAVE_ELEMS = 16 * FS_ROW; // complex slots per wave
constexpr int FS_LDS_BYTES = (FS_WAVES * FS_WAVE_ELEMS + 64) * 8; // data per wave + tw2 table

struct FsEstParams {
    const cf* in; long in_stride;   // [nchan][n] new items
    const cf* pend; int npend;      // [nchan][fftlen] pending partial vector (npend items valid)
    const cf* wtab;                 // [1024] W_1024^k
    int* maxpos; long maxpos_stride; // [nchan][nvec]
    int nvec, offset;               // freqest d_offset
};

template <class Ctx>
AISX_DI void fs_est_body(Ctx& cx, const FsEstParams& p)
{
    const int t = cx.tid();
    const int wave = t >> 6, l = t & 63, lane0 = l;
    const int c = cx.by();
    const int v0 = (cx.bx() * FS_WAVES + wave) * FS_VPW; // this wave's vectors: v0 .. v0 + FS_VPW - 1
    cf* lds = (cf*)cx.lds();
    cf* X = lds + wave * FS_WAVE_ELEMS;
    cf* T2 = lds + FS_WAVES * FS_WAVE_ELEMS; // W_64^{k2*n3}, index k2*4+n3
    if (t < 64)
        T2[t] = p.wtab[(16 * (t >> 2) * (t & 3)) & (FS_F - 1)];
    const cf* xin = p.in + (long)c * p.in_stride;
    const cf* pend = p.pend + (long)c * FS_F;
    cf tw1[16];
    tw1[0] = mk(1.f, 0.f);
#pragma unroll
    for (int k1 = 1; k1 < 16; k1++)
        tw1[k1] = p.wtab[(k1 * l) & (FS_F - 1)];
    // lane l = column (n2,n3) of vector v: s[l + 64 n1], n1 = 0 .. 15
    auto fetch = [&](int v, cf (&s)[16]) {
        const long i0 = (long)v * FS_F; // index of the vector's first item in pending ++ new
        // (all three tests are wave-uniform; one base pointer and constant offsets per case -- sixteen per-item
        // selects between two rows would keep thirty-two addresses alive across the loop)
        if (v >= p.nvec) {
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++)
                s[n1] = mk(0.f, 0.f);
        } else if (i0 >= p.npend) {
            const cf* b = xin + (i0 - p.npend) + l;
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++)
                s[n1] = b[64 * n1];
        } else if (i0 + FS_F <= p.npend) {
            const cf* b = pend + i0 + l;
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++)
                s[n1] = b[64 * n1];
        } else { // the vector the pending items end in
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                const long idx = i0 + l + 64 * n1;
                s[n1] = (idx < p.npend) ? pend[idx] : xin[idx - p.npend];
            }
        }
    };
    cf nxt[16];
    fetch(v0, nxt);
    cx.sync(); // (the W_64 table wave 0 wrote is in place for every wave; from here on a wave only meets its own rows)
#pragma nounroll
    for (int it = 0; it < FS_VPW; it++) {
    const int v = v0 + it;
    if (v >= p.nvec)
        break; // (the whole wave: v is wave-uniform)
    const bool live = true;
    // (the lane index of the body, opaque to the optimiser: otherwise every LDS address and index below -- a hundred
    // of them, all functions of the lane alone -- is hoisted out of this loop and kept in registers across it: 230 VGPRs)
    int l = lane0;
    cx.pin(l);
    cf x[16];
    // P1: x[n1] = s[l + 64 n1]^2
#pragma unroll
    for (int n1 = 0; n1 < 16; n1++)
        x[n1] = cmul_exact(nxt[n1], nxt[n1]); // multiply_cc of the stream with itself
#if FS_PREFETCH
    if (it + 1 < FS_VPW)
        fetch(v + 1, nxt); // (in flight while this vector is transformed)
#endif
    dft16<false>(cx, x);
#pragma unroll
    for (int k1 = 1; k1 < 16; k1++)
        x[k1] = cmul_fma(x[k1], tw1[k1]);
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++)
        X[k1 * FS_ROW + l] = x[k1];
    cx.wave_lds_sync();
    {
        const int k1 = l >> 2, n3 = l & 3;
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++)
            x[n2] = X[k1 * FS_ROW + n2 * 4 + n3];
        dft16<false>(cx, x);
#pragma unroll
        for (int k2 = 1; k2 < 16; k2++)
            x[k2] = cmul_fma(x[k2], T2[k2 * 4 + n3]);
#pragma unroll
        for (int k2 = 0; k2 < 16; k2++)
            X[k1 * FS_ROW + k2 * 4 + n3] = x[k2];
    }
    cx.wave_lds_sync();
    cf y[16];
#pragma unroll
    for (int h = 0; h < 4; h++) {
        const int q = l + 64 * h, k1 = q >> 4, k2 = q & 15;
        cf y0 = X[k1 * FS_ROW + k2 * 4 + 0], y1 = X[k1 * FS_ROW + k2 * 4 + 1];
        cf y2 = X[k1 * FS_ROW + k2 * 4 + 2], y3 = X[k1 * FS_ROW + k2 * 4 + 3];
        dft4<false>(cx, y0, y1, y2, y3);
        y[4 * h + 0] = y0;
        y[4 * h + 1] = y1;
        y[4 * h + 2] = y2;
        y[4 * h + 3] = y3;
    }
    cx.wave_lds_sync();
    // the spectrum in fft-shifted order: S(j) = X[(j + F/2) mod F]  <=>  j = (k + F/2) mod F.  Item j lives in
    // slot j + (j >> 4): neighbouring lanes write j's that differ by 16 (k = k1 + 16 k2 + 256 k3 with k2 = l & 15),
    // which in a dense array is one bank pair for all of them; with one slot skipped per 16 the stride is 17.
    // The readers (consecutive j) lose nothing.  1024 + 64 slots = FS_WAVE_ELEMS exactly.
    static_assert(FS_F + FS_F / 16 <= FS_WAVE_ELEMS, "the padded spectrum fits the wave's LDS");
    cf* S = X;
    auto sl = [](int j) { return j + (j >> 4); };
    {
        // value 4 h + k3 of this lane is frequency k = kb + 256 k3, kb = k1 + 16 k2 with k1 = (l >> 4) + 4 h, k2 = l & 15;
        // j = (k + 512) mod 1024 = kb + 256 ((k3 + 2) mod 4), and kb < 256 keeps j >> 4 = (kb >> 4) + 16 ((k3 + 2) mod 4):
        // slot = slot0 + 4 h (+ carry-free: k1 < 16) + 272 ((k3 + 2) mod 4)
        const int kb0 = (l >> 4) + 16 * (l & 15);
        const int slot0 = kb0 + (kb0 >> 4);
#pragma unroll
        for (int h = 0; h < 4; h++)
#pragma unroll
            for (int k3 = 0; k3 < 4; k3++)
                S[slot0 + 4 * h + 272 * ((k3 + 2) & 3)] = y[4 * h + k3];
    }
    cx.wave_lds_sync();
    // freqest search (lib/freqest_impl.cc:74-83): the first strict maximum of e[j] = |S[j]| + |S[j + offset]|,
    // each magnitude glibc's hypotf (cabs_f: double products, a double square root -- ~45 issue slots).
    // Round 5: that exact form is evaluated only where it can matter.  Pass 1 scores every j in single
    // precision (two products, a sum, v_sqrt_f32: within 3e-7 of the exact e[j]) and takes the wave's
    // maximum E; pass 2 evaluates the exact e[j] for the j whose score is within 4e-6 of E -- every other
    // j is below the exact maximum by more than both errors together and can neither be it nor tie it.
    // The result is the exact search's, bit for bit.  Scores the bound does not cover (E below 1e-10:
    // squares that underflow; above 1e18: squares that overflow; all-zero vectors) take pass 2 for every j.
    const int span = FS_F - p.offset;
    constexpr int NJ = FS_F / 64;
    float sc[NJ];
    float emax = 0.f;
#pragma unroll
    for (int i = 0; i < NJ; i++) {
        const int j = l + 64 * i;
        sc[i] = -1.f;
        if (j < span) {
            const cf a0 = S[sl(j)], a1 = S[sl(j + p.offset)];
            sc[i] = cx.sqrt_approx(a0.re * a0.re + a0.im * a0.im) + cx.sqrt_approx(a1.re * a1.re + a1.im * a1.im);
            emax = fmaxf(emax, sc[i]); // (a NaN score is skipped here as the exact search skips a NaN sum)
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        emax = fmaxf(emax, cx.shfl_xor_f32(emax, o));
    const bool covered = emax >= 1e-10f && emax <= 1e18f;
    const float thr = covered ? emax * (1.0f - 4e-6f) : -1.f;
    unsigned cand = 0; // slots of this lane that pass 2 evaluates
#pragma unroll
    for (int i = 0; i < NJ; i++)
        if (l + 64 * i < span && (sc[i] >= thr || !covered))
            cand |= 1u << i;
    float best = 0.f;
    int bestj = -1;
    // (one copy of the exact form, run as often as the busiest lane has candidates: once or twice as a rule;
    // slots are taken in rising order, so a lane meets its j in the sequential loop's order)
#pragma nounroll
    while (cx.ballot(cand != 0u) != 0ull) {
        if (cand != 0u) {
            const int i = __builtin_ctz(cand);
            cand &= cand - 1u;
            const int j = l + 64 * i;
            const float e = cabs_f(S[sl(j)]) + cabs_f(S[sl(j + p.offset)]);
            if (e > best) {
                best = e;
                bestj = j;
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ob = cx.shfl_xor_f32(best, o);
        const int oj = cx.shfl_xor_i32(bestj, o);
        // keep the larger energy; on a tie the smaller index (the sequential loop's first maximum)
        if (oj >= 0 && (bestj < 0 || ob > best || (ob == best && oj < bestj))) {
            best = ob;
            bestj = oj;
        }
    }
    if (live && l == 0)
        p.maxpos[(long)c * p.maxpos_stride + v] = (bestj >= 0) ? bestj + p.offset / 2 : -1;
    cx.wave_lds_sync(); // (the next vector's first pass overwrites the rows the search has just read)
#if !FS_PREFETCH
    if (it + 1 < FS_VPW)
        fetch(v + 1, nxt);
#endif
    } // vectors of this wave
}

// freqest::work on spectra the caller already transformed (lib/freqest_impl.cc:57-88):
// one wave per channel walks the vectors in order, so maxpos carries over exactly
// as in the reference.
struct FsFreqestParams {
    const cf* vecs; long vec_stride; // [nchan][nvec*fftlen], fft-shifted
    float* out; long out_stride;     // [nchan][nvec]
    int nvec, fftlen, offset;
    float binsize;
};

template <class Ctx>
AISX_DI void fs_freqest_body(Ctx& cx, const FsFreqestParams& p)
{
    const int l = cx.tid();
    const int c = cx.bx();
    const cf* X = p.vecs + (long)c * p.vec_stride;
    unsigned int maxpos = 0;
    const int span = p.fftlen - p.offset;
    for (int v = 0; v < p.nvec; v++) {
        const cf* V = X + (long)v * p.fftlen;
        float best = 0.f;
        int bestj = -1;
        for (int j = l; j < span; j += 64) {
            const float e = cabs_f(V[j]) + cabs_f(V[j + p.offset]);
            if (e > best) {
                best = e;
                bestj = j;
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ob = cx.shfl_xor_f32(best, o);
            const int oj = cx.shfl_xor_i32(bestj, o);
            if (oj >= 0 && (bestj < 0 || ob > best || (ob == best && oj < bestj))) {
                best = ob;
                bestj = oj;
            }
        }
        if (bestj >= 0)
            maxpos = (unsigned)(bestj + p.offset / 2);
        if (l == 0)
            p.out[(long)c * p.out_stride + v] = ((float)maxpos - (float)((unsigned)p.fftlen / 2)) * p.binsize / 2.0f;
    }
}

// ---------------------------------------------------------------------------
// The NCO phase walk on its own (fs_walk_body): the phases depend on the frequency estimates
// only, not on the samples, so the recurrence can run apart from the mixing -- one lane per
// channel, 64 channels per wave; the mixing then has no order in time any more and is done where
// the samples are read next (the AGC's load stage, k_agc.h).
// The walk is what fs_mix_body's wave 0 does, statement for statement (stale-maxpos rule
// lib/freqest_impl.cc:68 vs :74, f = (float(maxpos) - fftlen/2) * binsize / 2 (:84), [GR]
// frequency_modulator_fc's d_phase += k f; fmod wrap).  It leaves every FSW_CK-th phase in memory
// -- phi[c][FSW_CK j], 4 / FSW_CK bytes per sample instead of 4 -- and the per-vector increment d = k f:
// the reader walks the phases in between again (the same statement, aisx_common.h nco_phase_step).  A
// wave keeps FSW_BLK checkpoints of its 64 channels in LDS and flushes them row by row.
constexpr int FSW_T = 64;
constexpr int FSW_CK = NCO_CK; // samples per checkpoint (aisx_common.h)
// checkpoints per flush.  Small on purpose: 3 KB of LDS per wave -- the walk is meant to run beside the
// timing recovery (84 KB per workgroup) AND the correlator (72 KB) on the same CU, which leaves 4
#ifndef FSW_BLK
#define FSW_BLK 8
#endif
constexpr int FSW_PITCH = FSW_BLK + 4; // floats per LDS row (rows 16-byte aligned)
constexpr int FSW_LDS_BYTES = FSW_T * FSW_PITCH * 4;
static_assert(FS_F % (FSW_BLK * FSW_CK) == 0 && FSW_BLK % 4 == 0 && FSW_BLK <= 64, "whole 16-byte quads, whole blocks per vector");

struct FsWalkParams {
    int nchan;
    const int* maxpos; long maxpos_stride; // [nchan][nvec] from fs_est_body
    float* fhat; long fhat_stride;         // optional [nchan][nvec]
    const float* phase_in; float* phase_out; // [nchan] NCO phase (d_phase) before / after (may be the same array)
    float* phases; long phases_stride;     // [nchan][nvec * fftlen / FSW_CK] out: the phase of every FSW_CK-th item; stride a multiple of 4
    float* dvec; long dvec_stride;         // [nchan][nvec] out: the phase increment of each vector's items
    int nvec;
    float binsize, sensitivity;
};

template <class Ctx>
AISX_DI void fs_walk_body(Ctx& cx, const FsWalkParams& p)
{
    typedef float ph4 __attribute__((vector_size(16)));
    const int l = cx.tid();
    const int cbase = cx.bx() * FSW_T;
    const int c = cbase + l;
    const bool live = c < p.nchan;
    float* R = (float*)cx.lds(); // [64][FSW_PITCH]: row = channel of the wave, column = checkpoint of the block
    float* mine = R + l * FSW_PITCH;
    float ph = live ? p.phase_in[c] : 0.f;
    unsigned int maxpos = 0; // freqest_impl.cc:68 -- initialised once per work() call
    for (int v = 0; v < p.nvec; v++) {
        float d = 0.f;
        if (live) {
            const int mp = p.maxpos[(long)c * p.maxpos_stride + v];
            if (mp >= 0)
                maxpos = (unsigned)mp;
            const float f = ((float)maxpos - (float)(FS_F / 2)) * p.binsize / 2.0f; // :84
            if (p.fhat)
                p.fhat[(long)c * p.fhat_stride + v] = f;
            d = p.sensitivity * f;
            p.dvec[(long)c * p.dvec_stride + v] = d;
        }
        // with |d| < 2 pi the fmod of the wrap is a select (nco_wrap_small); one wave-uniform
        // decision per vector keeps the recurrence free of branches
        const bool small = cx.ballot(!(fabsf(d) < 6.0f)) == 0ull;
        for (int b = 0; b < FS_F / (FSW_BLK * FSW_CK); b++) {
            if (small) {
#pragma unroll 2
                for (int i = 0; i < FSW_BLK; i++) {
                    ph = nco_wrap_small(ph + d);
                    mine[i] = ph; // item FSW_CK j
#pragma unroll
                    for (int k = 1; k < FSW_CK; k++)
                        ph = nco_wrap_small(ph + d);
                }
            } else {
                for (int i = 0; i < FSW_BLK; i++) {
                    ph = nco_wrap(ph + d);
                    mine[i] = ph;
                    for (int k = 1; k < FSW_CK; k++)
                        ph = nco_wrap(ph + d);
                }
            }
            cx.wave_sync();
            // flush: QPR lanes x 16 bytes cover one channel's FSW_BLK checkpoints, 64 / QPR channels per pass
            constexpr int QPR = FSW_BLK / 4;
            const long col = (long)v * (FS_F / FSW_CK) + b * FSW_BLK + 4 * (l % QPR);
#pragma unroll 4
            for (int pass = 0; pass < QPR; pass++) {
                const int r = pass * (FSW_T / QPR) + l / QPR;
                const ph4 q = *(const ph4*)(R + r * FSW_PITCH + 4 * (l % QPR));
                if (cbase + r < p.nchan)
                    *(ph4*)(p.phases + (long)(cbase + r) * p.phases_stride + col) = q;
            }
            cx.wave_sync();
        }
    }
    if (live)
        p.phase_out[c] = ph;
}

// ---------------------------------------------------------------------------
constexpr int FSM_T = 512;             // wave 0 walks the NCO phases, waves 1..7 mix
constexpr int FSM_CPW = 16;            // channels per workgroup (256 workgroups at 4096 channels: one per CU)
constexpr int FSM_MIXW = FSM_T / 64 - 1;
constexpr int FSM_CH = 128;            // samples per chunk
constexpr int FSM_PITCH = FSM_CH + 4;  // floats per channel row in LDS (rows 16-byte aligned: the walk stores four phases at a time)
constexpr int FSM_UNITS = FSM_CPW * (FSM_CH / 64);            // (channel, 64-sample half) units per chunk
constexpr int FSM_UPW = (FSM_UNITS + FSM_MIXW - 1) / FSM_MIXW; // units per mixing wave
constexpr int FSM_LDS_BYTES = 2 * FSM_CPW * FSM_PITCH * 4 + NCO_TAB_FLOATS * 4; // two phase buffers + the NCO's sine table

struct FsMixParams {
    int nchan;
    const cf* in; long in_stride;
    const cf* pend_in; cf* pend_out; int npend; // pending partial vector in / out
    int n;                                       // new items per channel
    cf* out; long out_stride;
    const int* maxpos; long maxpos_stride;
    float* fhat; long fhat_stride;               // optional [nchan][nvec]
    float* phase;                                // [nchan] NCO phase (d_phase)
    int nvec;
    float binsize, sensitivity;
    const float* sintab;                         // gr::fxpt's sine table (NCO_TAB_FLOATS floats)
};

template <class Ctx>
AISX_DI void fs_mix_body(Ctx& cx, const FsMixParams& p)
{
    const int t = cx.tid();
    const int wave = t >> 6, l = t & 63;
    const int cbase = cx.bx() * FSM_CPW;
    float* PH = (float*)cx.lds(); // [2][FSM_CPW][FSM_PITCH]
    float* ST = PH + 2 * FSM_CPW * FSM_PITCH; // the NCO's sine table (first used behind the first barrier)
    {
        typedef float f4 __attribute__((vector_size(16)));
        for (int i = t; i < NCO_TAB_FLOATS / 4; i += FSM_T)
            ((f4*)ST)[i] = ((const f4*)p.sintab)[i];
    }
    // wave 0: lane l < FSM_CPW walks channel cbase + l
    const int myc = cbase + l;
    const bool mylive = (wave == 0) && (l < FSM_CPW) && (myc < p.nchan);
    float ph = mylive ? p.phase[myc] : 0.f;
    unsigned int maxpos = 0; // freqest_impl.cc:68 -- initialised once per work() call
    float d = 0.f;
    const int total = p.nvec * FS_F;
    const int nchunks = total / FSM_CH;
    // software pipeline: wave 0 produces the phases of chunk k while waves 1..7 mix chunk k-1
    cf nxt[FSM_UPW];
#pragma unroll
    for (int q = 0; q < FSM_UPW; q++)
        nxt[q] = mk(0.f, 0.f);
    auto fetch = [&](long k0) {
#pragma unroll
        for (int q = 0; q < FSM_UPW; q++) {
            const int u = (wave - 1) + FSM_MIXW * q;
            const int r = u % FSM_CPW, c = cbase + r;
            const long idx = k0 + (u / FSM_CPW) * 64 + l;
            nxt[q] = mk(0.f, 0.f);
            if (u < FSM_UNITS && c < p.nchan)
                nxt[q] = (idx < p.npend) ? p.pend_in[(long)c * FS_F + idx] : p.in[(long)c * p.in_stride + idx - p.npend];
        }
    };
    for (int k = 0; k <= nchunks; k++) {
        if (wave == 0) {
            if (mylive && k < nchunks) {
                const int k0 = k * FSM_CH;
                float* dst = PH + (k & 1) * FSM_CPW * FSM_PITCH + l * FSM_PITCH;
                if ((k0 & (FS_F - 1)) == 0) { // a new vector starts
                    const int v = k0 / FS_F;
                    const int mp = p.maxpos[(long)myc * p.maxpos_stride + v];
                    if (mp >= 0)
                        maxpos = (unsigned)mp;
                    // out[i] = (float(maxpos) - fftlen/2) * d_binsize/2   (:84)
                    const float f = ((float)maxpos - (float)(FS_F / 2)) * p.binsize / 2.0f;
                    if (p.fhat)
                        p.fhat[(long)myc * p.fhat_stride + v] = f;
                    d = p.sensitivity * f;
                }
                // [GR] frequency_modulator_fc_impl::work: d_phase += sensitivity * in[i], then
                // the fmod wrap.  With |d| < 2 pi the argument of the wrap stays below 4 pi in
                // magnitude and fmod is a select (nco_wrap_small): no branch in the recurrence.
                if (fabsf(d) < 6.0f) {
                    // (a lone wave pays ~18 cycles of issue per LDS instruction: one 128-bit
                    // store per four phases)
                    typedef float ph4 __attribute__((vector_size(16)));
#pragma unroll 2
                    for (int i = 0; i < FSM_CH; i += 4) {
                        ph4 v;
                        ph = nco_wrap_small(ph + d);
                        v[0] = ph;
                        ph = nco_wrap_small(ph + d);
                        v[1] = ph;
                        ph = nco_wrap_small(ph + d);
                        v[2] = ph;
                        ph = nco_wrap_small(ph + d);
                        v[3] = ph;
                        *(ph4*)(dst + i) = v;
                    }
                } else {
                    for (int i = 0; i < FSM_CH; i++) {
                        ph = nco_wrap(ph + d);
                        dst[i] = ph;
                    }
                }
            }
        } else {
            // the items of chunk k are fetched now and mixed in the next trip (their latency
            // hides behind this trip's sin/cos and the barrier)
            cf cur[FSM_UPW];
#pragma unroll
            for (int q = 0; q < FSM_UPW; q++)
                cur[q] = nxt[q];
            if (k < nchunks)
                fetch((long)k * FSM_CH);
            if (k > 0) {
                const long k0 = (long)(k - 1) * FSM_CH;
                const float* src = PH + ((k - 1) & 1) * FSM_CPW * FSM_PITCH;
#pragma unroll
                for (int q = 0; q < FSM_UPW; q++) {
                    const int u = (wave - 1) + FSM_MIXW * q;
                    const int r = u % FSM_CPW, c = cbase + r;
                    const int h = u / FSM_CPW;
                    if (u < FSM_UNITS && c < p.nchan) {
                        float sn, cs;
                        nco_sincos(src[r * FSM_PITCH + h * 64 + l], ST, &sn, &cs); // [GR] gr::fxpt::sincos
                        p.out[(long)c * p.out_stride + k0 + h * 64 + l] = cmul_exact(cur[q], mk(cs, sn));
                    }
                }
            }
        }
        cx.sync();
    }
    if (mylive)
        p.phase[myc] = ph;
    // keep the trailing partial vector (stream_to_vector's pending items)
    const int rem = p.npend + p.n - total;
    for (int r = wave; r < FSM_CPW; r += FSM_T / 64) {
        const int c = cbase + r;
        if (c < p.nchan)
            for (int i = l; i < rem; i += 64) {
                const long idx = (long)total + i;
                p.pend_out[(long)c * FS_F + i] =
                    (idx < p.npend) ? p.pend_in[(long)c * FS_F + idx] : p.in[(long)c * p.in_stride + idx - p.npend];
            }
    }
}

} // namespace aisx
