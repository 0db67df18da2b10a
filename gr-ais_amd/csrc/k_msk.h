// k_msk.h -- msk_timing_recovery_cc (reference: lib/msk_timing_recovery_cc_impl.cc
// :107-206), and the NRZI bit tail of python/ais_demod.py:48-52 + lib/invert_impl.cc
// :62-64 as a second, fully parallel kernel over the symbols the first one wrote.
//
// The loop is a strict recurrence through (mu, omega, iidx): the only parallelism
// is across channels.  A workgroup is msk_waves(LPW) waves of LPW channels each (LPW = 8
// by default: four waves, one per SIMD, 32 channels), each wave on its own; inside a wave
// all 64 lanes stay enabled, lane i carrying the channel of lane i % LPW.  Every channel keeps a ring
// of its last 256 samples in LDS, stored slot-major (ring[slot][channel of the wave]): an
// 8-tap read never conflicts however far the channels drift apart.  Chunk t = new samples
// [64t, 64t+64) of every channel is fetched into registers one chunk ahead (16-byte loads,
// the 64 / LPW lanes of a channel each taking a share) and lands in the rings as soon as
// no lane still reads the slots it overwrites.  Symbols are stored straight from the loop.
//
// A lone wave issues one instruction every ~2.3 ns whatever the dependencies and however
// few lanes are enabled (tools/ubench), so the loop is written for instruction count:
//  * the reference's iteration comes in two kinds, d_div even (emit a symbol) and
//    d_div odd (run the loop filter).  While no lane is near an event, (even, odd) pairs
//    run in lock step in a counted loop with no test inside: the number of pairs every
//    lane can run follows from a per-lane bound fast_lim on iidx (end of this
//    general_work call, data horizon, next time_est tag) and the proven maximum advance
//    of a pair; the wave takes the minimum.  In a run two lanes share a channel's pair:
//    one does the even iteration, the other the odd one at the same time (the even one
//    has no feedback into mu), the squares cross over by a row swap;
//  * lanes at an event run the reference's loop head in its order (events()) behind
//    wave-uniform branches, then even and odd iterations as masked steps; parity and
//    "parked" are 64-bit wave masks in scalar registers.  A tag reset can shift a lane
//    by one iteration, it rejoins the lock step on the next pass;
//  * the time_est tags come compacted (tagprep_body) and are queued in LDS, so a firing
//    tag costs an LDS read;
//  * the bit tail (quadrature demod, slicer, differential decoder, invert) has no
//    feedback into the loop and runs afterwards over all symbols in parallel
//    (bittail_body).
// All arithmetic is the reference's float/double sequence, unfused: bit-identical
// to the CPU restatement.
#pragma once
#include "aisx_common.h"

namespace aisx {

enum {
    MSK_ST_INTERP_RANGE = 1,
    MSK_ST_CARRY_OVERFLOW = 2,
    MSK_ST_TAGCARRY_OVERFLOW = 4,
    MSK_ST_OUT_FULL = 8,
    MSK_ST_TAGS_TRUNCATED = 16 // the producer's tag list was longer than its buffer: tags are missing
};
constexpr int MSK_CTN_TRUNC = 0x40000000; // flag bit in ct_n: tagprep_body saw a truncated list
constexpr int MSK_CTN_WILD = 0x20000000;  // ... a time_est value outside [-1, 1] (corr_est's centre of mass never is)
constexpr int MSK_TAG_TRIPS = 6;          // pairs per run that handles tag resets in line (a burst gives 3-4 tags on consecutive pairs)

constexpr int MSK_T = 64;    // lanes of a wave
constexpr int MSK_RING = 256;   // slots per lane (power of two)
constexpr int MSK_SLOTS = MSK_RING + 8; // + 8 mirror slots: an 8-tap read never wraps
constexpr int MSK_CHUNK = 64;   // samples per chunk
constexpr int MSK_OFF = 192;    // ring slot of new-sample index s is (s + MSK_OFF) & 255
constexpr int MSK_CARRY_MAX = 128;
constexpr int MSK_PAIRS_MAX = 16; // pairs of iterations per check-free run (a power of two)
constexpr int MSK_TAPS_PITCH = 12; // floats per table row in LDS: 16-byte aligned rows, two 128-bit reads per FIR
constexpr int msk_lds_ring(int lpw) { return MSK_SLOTS * lpw * 8; } // rings: [MSK_SLOTS][lpw] samples
constexpr int MSK_ZERO_ROW = 129; // an all-zero tap row: where an out-of-range mu lands
constexpr int MSK_LDS_MMSE = ((130 * MSK_TAPS_PITCH * 4 + 511) / 512) * 512; // whole slot rows: folds into ds offsets
constexpr int MSK_TAGQ = 36;       // time_est tags queued per lane
// a workgroup is msk_waves(lpw) waves with lpw channels each: 64 channels for lpw >= 16, four
// waves (one per SIMD) of 8 or 4 channels below; every wave has its own rings and tag queue, the
// tap table behind them is shared
// (measured, round 2: 4 channels per wave with EIGHT waves per workgroup -- two per SIMD on the
// same 128 CUs -- is slower than 8 channels on four waves: 7.6 against 6.55 ms per step of the
// whole flowgraph, 6.3 against 4.45 on corr_est -> msk; -DMSK_WAVES_LPW4=8 rebuilds that variant)
#ifndef MSK_WAVES_LPW4
#define MSK_WAVES_LPW4 4
#endif
constexpr int msk_waves(int lpw) { return lpw >= 16 ? 64 / lpw : (lpw == 4 ? MSK_WAVES_LPW4 : 4); }
constexpr int msk_wg_channels(int lpw) { return msk_waves(lpw) * lpw; }
// Symbols are staged in LDS and leave in 16-byte stores, 2 * (64 / lpw) symbols of a channel at a
// time: MSK_STAGE slots per channel, [slot][lpw] like the rings, all waves' stages at the very
// start of the workgroup's LDS (each aligned to its size: a slot address is one v_and_or).  A
// run of the loop adds at most MSK_PAIRS_MAX + 1 symbols to fewer than 16 left by the flush.
constexpr int MSK_STAGE = 32;
static_assert(MSK_STAGE >= 15 + MSK_PAIRS_MAX + 1, "a lock-step run and a general step between two flushes");
// (builds with 16 or more channels per wave fill the LDS with rings: they store symbol by symbol)
constexpr bool msk_staged(int lpw) { return lpw <= 8; }
constexpr int msk_lds_stage(int lpw) { return msk_staged(lpw) ? MSK_STAGE * lpw * 8 : 0; }
constexpr int msk_lds_ringoff(int lpw) { return msk_waves(lpw) * msk_lds_stage(lpw); } // first wave's rings
// a wave's region: rings, tag queue, one spare slot row (where the lanes that run odd iterations "stage")
constexpr int msk_lds_wave(int lpw) { return msk_lds_ring(lpw) + MSK_TAGQ * lpw * 8 + lpw * 8; }
constexpr int msk_lds_taboff(int lpw) { return msk_lds_ringoff(lpw) + msk_waves(lpw) * msk_lds_wave(lpw); }
constexpr int msk_lds_bytes(int lpw) { return msk_lds_taboff(lpw) + MSK_LDS_MMSE; }
static_assert(msk_lds_bytes(4) <= 160 * 1024 && msk_lds_bytes(8) <= 160 * 1024 && msk_lds_bytes(16) <= 160 * 1024 &&
              msk_lds_bytes(32) <= 160 * 1024 && msk_lds_bytes(64) <= 160 * 1024, "a workgroup's LDS");
// (8 channels per wave: 92160 bytes, which leaves room for one 71680-byte workgroup of the correlator on the same CU)
static_assert(msk_lds_bytes(8) + 71680 <= 160 * 1024, "timing recovery + one correlator workgroup per CU");
constexpr int BT_T = 256;          // bit tail: threads per workgroup
constexpr int BT_SEG = BT_T * 8;   // symbols per workgroup

// ---- restart points of the time-parallel recovery (k_mskp.h), as this kernel meets them when it
// runs as the JOIN of that scheme (MskParams::ff): see k_mskp.h for what they are
constexpr int MSKP_SMAX = 64; // restart points per channel at most
enum { MSKP_KIND_NONE = 0, MSKP_KIND_NEXT = 1, MSKP_KIND_HANDOFF = 2 };
struct mskp_rst { // restart point k of a channel: tags jA and jA + 1 of the channel's list of new time_est tags
    int jA;
    int relA, relB; // row offsets of the two tags
    int q0, cap;    // the unit's slots in the channel's staging row (cap 0: no unit runs from here)
    int pad[3];
};
struct mskp_snap { // the loop at the top of an iteration, before the tag test (:138-140)
    int a;          // iidx, as a row offset
    float mu, omega;
    int div;
    cf y;           // d_dly_conj_1 = d_dly_conj_2 (:194-195)
    cf nl;          // d_dly_diff_1
    int cur;        // index of the front tag in the new-tag list
    int cnt;        // symbols emitted
};
struct mskp_res { // what a unit leaves behind
    mskp_snap end;
    cf ay, anl;     // the delay registers it assumed at its restart point
    int kind;       // MSKP_KIND_*: ended where the next restart point's tag B is about to fire / somewhere else
    int status;
};
struct mskp_piece {
    int out0, src0, cnt; // symbols [out0, out0 + cnt) of the output row = staging row [src0, src0 + cnt)
};
AISX_HD bool mskp_same_bits(cf a, cf b)
{
    unsigned ar, ai, br, bi;
    __builtin_memcpy(&ar, &a.re, 4);
    __builtin_memcpy(&ai, &a.im, 4);
    __builtin_memcpy(&br, &b.re, 4);
    __builtin_memcpy(&bi, &b.im, 4);
    return ar == br && ai == bi;
}

// a time_est tag as the timing-recovery kernel wants it: offset relative to the channel's
// nitems_read at the start of the call, value narrowed to the float the loop uses
struct msk_ctag { int rel; float val; };

// two consecutive samples of a row; rows are only 8-byte aligned
struct __attribute__((packed, aligned(8))) cf_pair { cf a, b; };

struct MskParams {
    int nchan;
    // loop constants (set_sps / set_gain / set_limit, :69-96)
    float d_sps, gain, gain_omega, limit;
    int osps;
    // per-channel loop state
    float* mu; float* omega; int* div;
    cf* dly1; cf* dly2; cf* diff1;
    unsigned long long* nread; // nitems_read(0)
    // input: stream mode = carry (pre-item + pending items) followed by n new items
    const cf* in; long in_stride; int n;
    const cf* carry_in; cf* carry_out; const int* carry_len_in; int* carry_len_out; int carry_cap;
    // GNU Radio mode (stream_mode == 0): explicit ninput/noutput, nothing carried but the pre-item
    int stream_mode; int gr_ninput; int gr_noutput;
    // tags: the time_est tags of (carried ones + this call's), compacted by tagprep_body
    const msk_ctag* ct; const int* ct_n; int ct_cap;
    // tags still pending at the end of the call, for the next one
    tag_rec* ctag_out; int* ctag_n_out; int ctag_cap;
    // outputs (syms is mandatory; rows of one wave must lie within 4 GiB: out_stride < 2^23)
    cf* syms; float* err; float* mu_out; long out_stride; int out_cap;
    int* produced; int* consumed; int* status;
    const float* mmse; // [129][8]
    int lds_tab_off;   // = msk_lds_taboff(lpw)
    // LDS bytes between the regions of consecutive waves, tag-queue entries between consecutive
    // queue slots, and whether every lane has a queue column of its own.  On the device the NQ
    // lanes of a channel share one column (they run in lock step and write the same values:
    // stride msk_lds_wave(lpw), lpw, 0); the CPU lane model, whose lanes are free-running
    // threads, gives each lane its own (stride msk_lds_ring(lpw) + MSK_TAGQ * 64 * 8, 64, 1).
    int lds_wave_stride, tq_stride, tq_private;
    int lds_ring_off;  // = msk_lds_ringoff(lpw): the waves' regions start behind the symbol stages
    int sym_al16;      // every output row starts 16-byte aligned (syms pointer and out_stride both even in items)
    int lpw;           // channels per wave, = the build's LPW: 4, 8, 16, 32 or 64
    int inline_tags;   // tag resets inside the lock-step runs (0: every tag through the general steps)
    int max_noutput;   // gr::block::set_max_noutput_items(): output items a general_work call is offered at most (0: what fits)
    // ---- this kernel as the JOIN of the time-parallel recovery (ff != 0; stream mode, osps 1, no err / mu ports).
    // Where the front tag is tag B of restart point k, about to reset the loop, and the two delay registers
    // equal what unit k assumed, unit k's run IS this loop's: the lane takes its symbol count and end
    // state, and the following units' too while each ended on the next one's assumption (k_mskp.h).
    int ff;
    const int* nrst; const mskp_rst* rst; const mskp_res* res;
    mskp_piece* pieces; int* npieces;
    const int* ct_nc;  // carried tags in front of the new ones in `ct` (-1: a new tag was dropped, no fast-forward)
};

// quadrature_demod_cf(pi/2) -> binary_slicer_fb -> diff_decoder_bb(2) -> invert over the
// symbols of one msk call; the previous symbol / previous sliced bit are the only state
struct BitTailParams {
    int nchan;
    const cf* syms; long sym_stride;
    const int* produced;
    unsigned char* bits; long bit_stride;
    const cf* prev_sym_in; const unsigned char* prev_bit_in; // state before this call
    cf* prev_sym_out; unsigned char* prev_bit_out;           // state after it (a different buffer)
    const float* atan_tab;
};

AISX_HD int msk_forecast(float d_sps, int noutput_items)
{
    return (int)ceil((noutput_items * d_sps * 2) + 3.0 * d_sps + 8u);
}

#ifdef MSK_EMU_STATS
extern long msk_stats[8];
#endif
// AUX: the err / mu output ports are connected (:187-189).  OSPS2: osps == 2 (:186).
// LPW: channels (active lanes) per wave; the LDS layouts are [.][LPW], so a build with few
// channels per wave leaves most of the CU's LDS to whatever else runs there.
// FFT: the build that can run as the join of the time-parallel recovery (MskParams::ff; AUX and OSPS2 off).
template <class Ctx, bool AUX, bool OSPS2, int LPW, bool FFT = false>
AISX_DI void msk_body(Ctx& cx, const MskParams& p)
{
    static_assert(!FFT || (!AUX && !OSPS2), "the join build has no err / mu ports and osps == 1");
    constexpr int SLOT_B = LPW * 8;                                   // bytes per ring slot row
    constexpr int SLOT_SH = LPW == 64 ? 9 : (LPW == 32 ? 8 : (LPW == 16 ? 7 : (LPW == 8 ? 6 : 5))); // log2(SLOT_B)
    static_assert(LPW == 4 || LPW == 8 || LPW == 16 || LPW == 32 || LPW == 64, "channels per wave");
    typedef unsigned long long u64;
    // A workgroup is msk_waves(LPW) waves of LPW channels, every wave on its own (its own
    // SIMD, rings, tag queue, pace).  Few channels per wave: events of other lanes stall a
    // lane less, LDS returns fewer bytes per instruction, a chunk load touches fewer lines.
    static_assert(64 % LPW == 0, "whole waves");
    // All 64 lanes stay: lane i runs the channel of lane i % LPW (identical arithmetic, identical
    // stores to identical addresses -- a wave64 instruction costs the same with 16 or 64 lanes
    // enabled), and when a chunk is fetched and landed lane i handles share i / LPW of it, so a
    // chunk costs 64 / LPW times fewer instructions than with LPW lanes at work.
    constexpr int NQ = 64 / LPW;          // lanes per channel = shares of a chunk
    constexpr int QS = MSK_CHUNK / NQ;    // samples of a chunk per lane
    const int wv = cx.tid() >> 6;
    const int l = cx.tid() & (LPW - 1);   // the channel (ring column, tag queue column) of this lane
    const int q = (cx.tid() & 63) / LPW;  // its share of a chunk
    const bool owner = q == 0;            // the lane that writes the channel's state back
    const int cbase = cx.bx() * msk_wg_channels(LPW) + wv * LPW;
    const int c = cbase + l;
    const bool live = c < p.nchan;
    const int cc = live ? c : (p.nchan - 1); // dead lanes mirror the last channel read-only

    char* const lds0 = cx.lds();
    char* const lds = lds0 + p.lds_ring_off + wv * p.lds_wave_stride; // this wave's rings [MSK_SLOTS][LPW] and tag queue
    cf* ring = (cf*)lds;
    // [130][MSK_TAPS_PITCH], one per workgroup, behind the waves' regions; the offset comes in
    // as a kernel argument so that it sits in a scalar register and a row address is one
    // multiply-add
    float* mm = (float*)(lds0 + p.lds_tab_off);
    cf* myring = ring + l;                     // slot k of this lane: myring[k * LPW]

    for (int i = cx.tid(); i < 129 * 8; i += cx.nthreads())
        mm[(i >> 3) * MSK_TAPS_PITCH + (i & 7)] = p.mmse[i];
    if (cx.tid() < MSK_TAPS_PITCH)
        mm[MSK_ZERO_ROW * MSK_TAPS_PITCH + cx.tid()] = 0.f;
    if (cbase >= p.nchan) { // a wave with no channel at all (ragged last workgroup)
        cx.sync();
        return;
    }

    const float d_sps = p.d_sps;
    float d_mu = p.mu[cc], d_omega = p.omega[cc];
    int d_div = p.div[cc];
    // d_dly_conj_1 and d_dly_conj_2 both hold the previous interpolated sample (:194-195);
    // the loop only ever uses the square of it
    cf last_interp = p.dly1[cc];
    cf d_dly_diff_1 = p.diff1[cc];
    cf prev_sq = cmul_exact(p.dly2[cc], p.dly2[cc]);
    const unsigned long long R = p.nread[cc];
    int status = 0;
    const int n = p.n;

    // items on offer: logical index q in [-1, navail): q = -1 the item before
    // nitems_read, then the `pending` carried items, then the n new ones.  The new
    // sample index is s = q - pending; ring slot of s is (s + MSK_OFF) & 255.
    const cf* cin = p.carry_in + (long)cc * p.carry_cap;
    int pending = p.carry_len_in[cc];
    if (pending > MSK_CARRY_MAX)
        pending = MSK_CARRY_MAX;
    const int navail = pending + n;
    for (int q = -1; q < pending; q++) {
        const int slot = (q - pending + MSK_OFF) & (MSK_RING - 1);
        myring[slot * LPW] = cin[q + 1];
        if (slot < 8)
            myring[(MSK_RING + slot) * LPW] = cin[q + 1];
    }

    // the time_est tags visible to this call (carried ones first), already compacted
    const msk_ctag* ctl = p.ct + (long)cc * p.ct_cap;
    int ntot = p.ct_n[cc];
    if (ntot & MSK_CTN_TRUNC)
        status |= MSK_ST_TAGS_TRUNCATED;
    // tag resets are handled inside the lock-step runs only when every value is a timing offset
    // within one sample (then mu stays inside the interpolator's table after a reset)
    const bool tame_tags = p.inline_tags && cx.ballot((ntot & MSK_CTN_WILD) != 0) == 0ull;
    ntot &= MSK_CTN_WILD - 1;
    if (ntot > p.ct_cap)
        ntot = p.ct_cap;
    // They are queued in LDS: the loop must not pay global-memory latency when a tag fires.
    // MSK_TAGQ entries per lane, entry k of lane l at tq[k * LPW + l]; a longer list is queued
    // in instalments.
    typedef msk_ctag tq_ent;
    tq_ent* const tq = (tq_ent*)(lds + msk_lds_ring(LPW)) + (p.tq_private ? (cx.tid() & 63) : l);
    const int TQS = p.tq_stride;
    const int TQ_NONE = 0x7fffffff;
    int gq = 0;            // tags of the list looked at so far
    int qhead = 0, qn = 0; // queue: entries [0, qn), front at qhead
    auto tq_fill = [&]() {
        // (keeps the entry popped last: its offset may still be >= nitems_read at the end)
        if (qn > 0) {
            tq[0] = tq[(qn - 1) * TQS];
            qn = 1;
        }
        qhead = qn;
        while (qn < MSK_TAGQ && gq < ntot) {
            tq_ent t[4];
#pragma unroll
            for (int j = 0; j < 4; j++)
                t[j] = ctl[gq + j < ntot ? gq + j : ntot - 1]; // four loads in flight
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (qn < MSK_TAGQ && gq < ntot) {
                    gq++;
                    tq[qn * TQS] = t[j];
                    qn++;
                }
            }
        }
    };
    // the front of the queue is kept in registers
    int fr_rel = TQ_NONE; // offset of the front tag relative to R
    float nt_val = 0.f;
    int nt_rel = 0x7fffffff; // ... relative to this general_work call's nitems_read, if in its range
    auto tq_front = [&]() {
        if (qhead >= qn && gq < ntot)
            tq_fill();
        if (qhead < qn) {
            const tq_ent e = tq[qhead * TQS];
            fr_rel = e.rel;
            nt_val = e.val;
        } else {
            fr_rel = TQ_NONE;
        }
    };
    auto tq_pop = [&]() {
        qhead++;
        tq_front();
    };
    tq_fill();

    // ---- JOIN of the time-parallel recovery: the next restart point this lane may meet
    constexpr bool FF = FFT; // (the launcher picks the build by p.ff)
    // (FF: what a channel was when its step ended, see the main loop)
    bool fin_saved = false;
    float fin_mu = 0.f, fin_omega = 0.f;
    int fin_div = 0, fin_status = 0;
    cf fin_interp = mk(0.f, 0.f), fin_diff = mk(0.f, 0.f);
    unsigned fin_worst = 0;
    int ffK = 0, ffnck = 0, cand = 0, candj = 0x7fffffff, ffnp = 0;
    int voff = 0; // new-sample index = ring index + voff: behind a fast-forward the ring carries on elsewhere in the row
    const mskp_rst* const ffrs = FF ? p.rst + (long)cc * MSKP_SMAX : nullptr;
    const mskp_res* const ffrr = FF ? p.res + (long)cc * MSKP_SMAX : nullptr;
    if (FF) {
        ffnck = p.ct_nc[cc];
        ffK = ffnck >= 0 ? p.nrst[cc] : 0;
        candj = ffK > 0 ? ffrs[0].jA + 1 + ffnck : 0x7fffffff;
    }

    // output rows are addressed as (wave-uniform base) + (32-bit byte offset of this lane)
    char* const osym0 = (char*)(p.syms + (long)cbase * p.out_stride);
    char* const oerr0 = AUX && p.err ? (char*)(p.err + (long)cbase * p.out_stride) : nullptr;
    char* const omu0 = AUX && p.mu_out ? (char*)(p.mu_out + (long)cbase * p.out_stride) : nullptr;
    unsigned ob = (unsigned)((long)(cc - cbase) * p.out_stride) * 8u; // byte offset of the next symbol
    // ---- symbol stage (osps == 1, err / mu ports open, at most 8 channels per wave).  A global store per symbol costs the
    // recurrence more than its issue slots: the chunk fetch is awaited with s_waitcnt vmcnt(0),
    // which also waits for the store issued a moment ago.  Symbols go to LDS instead (slot =
    // symbol number mod MSK_STAGE) and leave FL at a time, each of the channel's lanes (up to eight)
    // storing two: whole 128-byte lines per channel, issued right after a chunk landed and
    // complete long before the next wait.
    constexpr bool STG = !AUX && !OSPS2 && msk_staged(LPW);
    constexpr unsigned FLQ = NQ < 8 ? NQ : 8;                     // lanes of a channel that store (two symbols each)
    constexpr unsigned FL = 2 * FLQ;                              // symbols per flush of a channel
    constexpr unsigned STG_MASK = (unsigned)((MSK_STAGE - 1) * SLOT_B);
    const unsigned obrow = ob;                                    // byte offset of the channel's output row
    const unsigned stg_real = (unsigned)(wv * msk_lds_stage(LPW) + l * 8);   // LDS byte offset of slot 0
    const unsigned stg_spare = (unsigned)(p.lds_ring_off + wv * p.lds_wave_stride + p.lds_wave_stride - SLOT_B + l * 8);
    unsigned so = 0;  // SLOT_B * (symbols of this call so far)
    unsigned nf = 0;  // symbols of this call already in the output row
    struct alignas(8) cf8s { float re, im; };
    auto stage_put = [&](unsigned mask, unsigned base, cf v) {
        cf8s t;
        t.re = v.re;
        t.im = v.im;
        *(cf8s*)(lds0 + ((so & mask) | base)) = t;
        so += (unsigned)SLOT_B;
    };
    auto stage_get = [&](unsigned k) -> cf {
        const cf8s t = *(const cf8s*)(lds0 + ((((k & (unsigned)(MSK_STAGE - 1)) << SLOT_SH)) | stg_real));
        return mk(t.re, t.im);
    };
    auto flush_syms = [&]() {
        if constexpr (STG) {
            if (cx.ballot((so >> SLOT_SH) - nf >= FL) == 0ull)
                return;
            cx.wave_sync(); // (lane model: the symbols the channel's other lanes staged are in place)
            while (cx.ballot((so >> SLOT_SH) - nf >= FL) != 0ull) {
                if ((so >> SLOT_SH) - nf >= FL) {
                    const unsigned k0 = nf + 2u * (unsigned)q;
                    if ((NQ <= 8 || (unsigned)q < FLQ) && !fin_saved) {
                        const cf a = stage_get(k0), b = stage_get(k0 + 1u);
                        cf* dst = (cf*)(osym0 + obrow) + k0;
                        if (p.sym_al16) {
                            st16(dst, a, b);
                        } else {
                            st8(dst, a);
                            st8(dst + 1, b);
                        }
                    }
                    nf += FL;
                }
            }
            cx.wave_sync(); // (lane model: read before the slots are staged again)
        }
    };

    // ---- "scheduler": one general_work() call after another (stream mode) ----
    int base = 0, ototal = 0;     // items consumed / produced by finished calls
    int iidx = 0, oidx = 0;       // of the call in progress
    int ninp = 0, noutput = 0;
    // dead lanes (c >= nchan) run as exact mirrors of the last channel, duplicate symbol stores
    // included (same address, same value), so that a ragged last wave stays in lock step
    bool done = false;
    auto setup_round = [&]() {
        int ninput;
        if (p.stream_mode) {
            ninput = (navail - base) - 1; // one look-ahead item is kept out of sight
            noutput = 0;
            if (ninput > 0) {
                noutput = (int)((ninput - 3.0 * d_sps - 8) / (2.0 * d_sps)) + 2;
                while (noutput > 0 && msk_forecast(d_sps, noutput) > ninput)
                    noutput--;
            }
            if (p.max_noutput > 0 && noutput > p.max_noutput)
                noutput = p.max_noutput;
            if (noutput > p.out_cap - ototal) {
                noutput = p.out_cap - ototal;
                status |= MSK_ST_OUT_FULL;
            }
        } else {
            ninput = p.gr_ninput;
            noutput = p.gr_noutput;
        }
        ninp = (int)(ninput - 3.0 * d_sps); // :119
        iidx = 0;
        oidx = 0;
        if (ninp <= 0 || noutput <= 0) {
            done = true;
            return;
        }
        // get_tags_in_range(nitems_read, nitems_read + ninp, "time_est") (:125-130)
        // (from the start of the queue: a tag an earlier call consumed is delivered again if
        // its offset is still >= nitems_read)
        qhead = 0;
        tq_front();
        while (fr_rel < base)
            tq_pop();
        nt_rel = (fr_rel != TQ_NONE && fr_rel - base < ninp) ? fr_rel - base : 0x7fffffff;
    };
    if (!done)
        setup_round();
    const int jump_margin = (int)ceilf(d_sps) + 1; // a tag may move iidx forward by < d_sps

    // chunks cover the new samples plus an 8-sample zero guard: the reference's loop
    // bound lets the interpolator look a few items past ninput_items when sps < 4;
    // here those items read as zero
    const int nchunks = (n + 8 + MSK_CHUNK - 1) / MSK_CHUNK;
    // chunk fetch: lane l walks its own channel row (64 samples = 512 contiguous bytes per
    // lane and chunk; the 16 lines involved stay in L1 across the 32 load instructions)
    cf r[QS]; // this lane's share of the chunk in flight
    const cf* myin = p.in + (long)cc * p.in_stride;
    auto issue_chunk = [&](int t) {
        const int s0 = t * MSK_CHUNK + q * QS + voff;
        // (FF: every lane's chunk lies somewhere else in its row)
        const bool inside = FF ? cx.ballot(!(s0 >= 0 && t * MSK_CHUNK + MSK_CHUNK + voff <= n)) == 0ull : (t * MSK_CHUNK + MSK_CHUNK <= n);
        if (inside) { // whole chunk inside the input: 16-byte loads, no predicates
            const cf_pair* src = (const cf_pair*)(myin + s0); // (dead lanes re-read the last channel's row)
#pragma unroll
            for (int k = 0; k < QS / 2; k++) {
                const cf_pair v = src[k];
                r[2 * k] = v.a;
                r[2 * k + 1] = v.b;
            }
        } else {
#pragma unroll
            for (int k = 0; k < QS; k++) {
                r[k] = mk(0.f, 0.f);
                if (s0 + k < n && s0 + k >= 0)
                    r[k] = myin[s0 + k];
            }
        }
    };
    auto land_chunk = [&](int t) {
        const int slot0 = (t * MSK_CHUNK + MSK_OFF) & (MSK_RING - 1); // multiple of 64
#pragma unroll
        for (int k = 0; k < QS; k++)
            myring[(slot0 + q * QS + k) * LPW] = r[k];
        if (slot0 == 0 && q * QS < 8) { // mirror the first 8 slots behind slot 255
#pragma unroll
            for (int k = 0; k < (QS < 8 ? QS : 8); k++)
                myring[(MSK_RING + q * QS + k) * LPW] = r[k];
        }
    };
    issue_chunk(0);
    land_chunk(0);
    int landed = 1; // chunks in the rings
    cx.sync();

    // sb = SLOT_B * (ring position of in[iidx]) = SLOT_B * (new-sample index + MSK_OFF), unmasked:
    // the byte offset of that slot row; it moves with iidx and is untouched by the end of a
    // general_work call (base += iidx, iidx = 0)
    int sb = (base + iidx - pending + MSK_OFF) * SLOT_B;
    // iterations with iidx < fast_lim (and oidx < noutput) need none of the event code
    int fast_lim = (int)0x80000000;
    int tag_trig = (int)0x80000000; // first iidx at which the front tag can fire (set by events())
    unsigned worst_imu = 0; // max over iterations of min(imu, MSK_ZERO_ROW)
    bool more = false;
    int loaded_s = 0;
    enum { EV_PARK = 0, EV_GO = 1, EV_OTHER_PARITY = 2 };
    const int TRIG_FORCED = (int)0x80000001;

    // the reference's loop head for one lane, in its order (:138-164)
    // how far a lane can run before the next event: the first iidx at which its front tag can
    // fire (tag_trig), the data horizon, the end of the general_work call
    auto bounds = [&](int spos) {
        tag_trig = 0x7fffffff;
        if (nt_rel != 0x7fffffff && iidx <= nt_rel) { // (a tag the loop stepped over stays in front for good)
            int i = nt_rel - jump_margin - 1;
            while (!((float)nt_rel < ((float)i + d_sps)))
                i++;
            tag_trig = i;
        }
        const int chunk_lim = more ? (loaded_s - 8 - jump_margin + MSK_OFF - (spos - iidx) + 1) : 0x7fffffff;
        fast_lim = ninp < tag_trig ? ninp : tag_trig;
        fast_lim = fast_lim < chunk_lim ? fast_lim : chunk_lim;
    };
    // FF: the lane stands where restart point `cand`'s tag B is about to reset the loop
    auto ff_walk = [&]() -> bool {
        bool moved = false;
        int dcnt = 0, a_new = 0, cur_new = 0;
        int oidx_w = oidx, nout_w = noutput, otot_w = ototal; // the general_work call in progress, as the walk goes
        bool crossed = false;
        cf yv = last_interp, nlv = d_dly_diff_1;
        float mu_new = 0.f, om_new = 0.f;
        int div_new = 0;
        const int cd = jump_margin - 1;
        for (;;) {
            const mskp_res r = ffrr[cand];
            const bool same = r.kind != MSKP_KIND_NONE && mskp_same_bits(r.ay, yv) && mskp_same_bits(r.anl, nlv);
            // (the unit ran blind to the general_work calls: k_mskp.h, mskp_body's walk, says when that is sound)
            bool clean = same && (r.end.a + 1 + cd <= base + ninp - pending) && ffnp < MSKP_SMAX;
            // (a unit that gave up -- at a stale tag, say -- with a call boundary inside: whether that tag
            // was dropped at the boundary or still blocks depends on where exactly the boundary fell)
            if (p.max_noutput > 0 && r.kind == MSKP_KIND_NEXT)
                clean = clean && msk_forecast(d_sps, p.max_noutput) <= (n - r.end.a) - 1 &&
                        (ototal + oidx + dcnt + r.end.cnt + p.max_noutput <= p.out_cap);
            else
                clean = clean && (oidx_w + r.end.cnt < nout_w);
            if (!clean) {
                cand++;
                candj = cand < ffK ? ffrs[cand].jA + 1 + ffnck : 0x7fffffff;
                break;
            }
            if (owner && live) {
                mskp_piece pc;
                pc.out0 = ototal + oidx + dcnt;
                pc.src0 = ffrs[cand].q0;
                pc.cnt = r.end.cnt;
                p.pieces[(long)cc * MSKP_SMAX + ffnp] = pc;
            }
            ffnp++;
            moved = true;
            dcnt += r.end.cnt;
            oidx_w += r.end.cnt;
            while (p.max_noutput > 0 && oidx_w >= nout_w) { // calls that began and ended inside the unit
                oidx_w -= nout_w;
                otot_w += nout_w;
                nout_w = p.max_noutput;
                crossed = true;
            }
            a_new = r.end.a;
            mu_new = r.end.mu;
            om_new = r.end.omega;
            div_new = r.end.div;
            yv = r.end.y;
            nlv = r.end.nl;
            cur_new = r.end.cur;
            status |= r.status;
            cand++;
            candj = cand < ffK ? ffrs[cand].jA + 1 + ffnck : 0x7fffffff;
            if (r.kind == MSKP_KIND_NEXT && cand < ffK)
                continue; // that unit ended where the next one's tag B is about to fire
            break;
        }
        if (!moved)
            return false;
        // what this lane has staged leaves now (the owner lane wrote those slots itself): the count jumps
        if constexpr (STG) {
            if (owner) {
                for (unsigned kk = nf; kk < (so >> SLOT_SH); kk++)
                    st8((cf*)(osym0 + obrow) + kk, stage_get(kk));
            }
            nf = (so >> SLOT_SH) + (unsigned)dcnt;
            so += (unsigned)dcnt << SLOT_SH;
        }
        ob += (unsigned)dcnt * 8u;
        d_mu = mu_new;
        d_omega = om_new;
        d_div = div_new;
        last_interp = yv;
        prev_sq = cmul_exact(yv, yv);
        d_dly_diff_1 = nlv;
        // the tags from the unit's front tag on
        gq = cur_new + ffnck;
        if (gq > ntot)
            gq = ntot;
        qn = 0;
        qhead = 0;
        tq_fill();
        const int q_new = a_new + pending; // logical index of in[iidx]
        tq_front();
        if (crossed) {
            // The call in progress began somewhere inside the units, at or before a_new and -- units stop at
            // the first stale tag -- at or before the front tag: nothing reads nitems_read before the call ends
            // except the tag range's lower end (:127), so any such place will do.
            ototal = otot_w;
            base = (fr_rel != TQ_NONE && fr_rel < q_new) ? fr_rel : q_new;
            setup_round(); // (noutput, ninp, the front tag)
        } else {
            nt_rel = (fr_rel != TQ_NONE && fr_rel - base < ninp) ? fr_rel - base : 0x7fffffff;
        }
        iidx = q_new - base;
        oidx = oidx_w;
        // The ring carries on at the unit's end: the slots of the chunk landed last are filled from there
        // right away (every lane of the channel writes the same 64 values: no exchange to wait for), the
        // chunk in flight is fetched again for this lane, and the lane goes on as if nothing had happened.
        const int v_new = loaded_s - MSK_CHUNK + 1; // in[iidx - 1] = the first slot of that chunk
        voff = a_new - v_new;
        sb = (v_new + MSK_OFF) * SLOT_B;
        {
            const int slot0 = (loaded_s - MSK_CHUNK + MSK_OFF) & (MSK_RING - 1); // multiple of 64
            // (on the device the NQ lanes of the channel each bring their share, loads first; the lane
            // model's lanes are free-running threads with no barrier here: each brings all 64)
            constexpr int SH = Ctx::wave_lds_coherent ? 1 : NQ;
            for (int sh = 0; sh < SH; sh++) {
                const int k0 = ((Ctx::wave_lds_coherent ? q : sh) * QS);
                cf t[QS];
#pragma unroll
                for (int k = 0; k < QS; k++) {
                    const int sr = loaded_s - MSK_CHUNK + k0 + k + voff;
                    t[k] = mk(0.f, 0.f);
                    if (sr >= 0 && sr < n)
                        t[k] = myin[sr];
                }
#pragma unroll
                for (int k = 0; k < QS; k++) {
                    myring[(slot0 + k0 + k) * LPW] = t[k];
                    if (slot0 == 0 && k0 + k < 8)
                        myring[(MSK_RING + k0 + k) * LPW] = t[k];
                }
            }
            if (more) { // (chunk `landed` is in flight, fetched for the old place)
                const int s0 = landed * MSK_CHUNK + q * QS + voff;
#pragma unroll
                for (int k = 0; k < QS; k++) {
                    r[k] = mk(0.f, 0.f);
                    if (s0 + k < n && s0 + k >= 0)
                        r[k] = myin[s0 + k];
                }
            }
        }
        tag_trig = (int)0x80000000;
        fast_lim = (int)0x80000000;
        return true;
    };
    auto events = [&](const int PAR) -> int {
        if (!(oidx < noutput && iidx < ninp)) { // this general_work() call is over (:138)
            base += iidx;                       // consume_each(iidx)
            ototal += oidx;
            // (a call that consumed nothing ends the step whatever it produced: with sps < 4 a tag
            // with a negative centre right at nitems_read can emit a symbol and leave iidx at 0
            // when one output fits -- called again with the same items it would do so for ever)
            const bool progress = iidx > 0;
            if (!p.stream_mode || !progress)
                done = true;
            else
                setup_round();
        }
        if (done)
            return EV_PARK;
        const int spos = sb >> SLOT_SH;
        // the next chunk has to land before this lane can go on (a tag is left for later too:
        // the iteration it resets must follow at once)
        const bool waiting = more && !(spos - MSK_OFF + 8 + jump_margin <= loaded_s);
        // a time_est tag lands in [iidx, iidx + d_sps) (:140-164)
        if (!waiting && (nt_rel >= iidx) && ((float)nt_rel < ((float)iidx + d_sps))) {
            if (FF) {
                const int fi = gq - (qn - qhead); // list index of the front tag
                while (cand < ffK && candj < fi) {
                    cand++;
                    candj = cand < ffK ? ffrs[cand].jA + 1 + ffnck : 0x7fffffff;
                }
                if (cand < ffK && fi == candj && ff_walk()) {
                    // (at the unit's end now, at the top of an iteration: from the loop head again)
                    bounds(sb >> SLOT_SH);
                    return EV_OTHER_PARITY;
                }
            }
            const float center = nt_val;
            if (center == center) { // not NaN (:144-147)
                const int old = iidx;
                d_mu = center;
                iidx = nt_rel;
                if (d_mu < 0) {
                    d_mu++;
                    iidx--;
                }
                sb += (iidx - old) * SLOT_B;
                d_div = 0;
                d_omega = d_sps;
                // (:160 d_dly_conj_2 = d_dly_conj_1: prev_sq already is the square of it)
            }
            tq_pop();
            nt_rel = (fr_rel != TQ_NONE && fr_rel - base < ninp) ? fr_rel - base : 0x7fffffff;
            // The reference runs this iteration whatever comes next (one tag per iteration,
            // :140).  Usually the bounds that hold after it can be set right away; if the next
            // event is already due, the lane gets one iteration and comes back here.
            bounds(sb >> SLOT_SH);
            if (fast_lim <= iidx) {
                tag_trig = TRIG_FORCED;
                fast_lim = iidx + 1;
            }
            return ((d_div & 1) == PAR) ? EV_GO : EV_OTHER_PARITY;
        }
        // nothing to do now: how far can this lane run before the next event?
        bounds(spos);
        if (waiting)
            return EV_PARK; // (the bound is re-armed when the chunk lands)
        return EV_GO;
    };

    // mmse_fir_interpolator_cc::interpolate(&in[..], mu) (:170) at ring byte position sbpos; an
    // imu outside [0, 128] (upstream throws std::runtime_error) reads the all-zero row
    auto tap_row = [&](float mu) -> unsigned {
        const unsigned imu = (unsigned)(int)rintf(mu * 128.0f);
        return imu < (unsigned)MSK_ZERO_ROW ? imu : (unsigned)MSK_ZERO_ROW;
    };
    // in two halves, so that the loads can be issued well before the sum
    auto fir_load = [&](unsigned row, int sbpos, cf* sv, float* tv) {
        typedef float tap4 __attribute__((vector_size(16)));
        const tap4* tp4 = (const tap4*)((const char*)mm + row * (unsigned)(MSK_TAPS_PITCH * 4)); // 16-byte aligned rows
        const tap4 tlo = tp4[0], thi = tp4[1];
        const float tp[8] = { tlo[0], tlo[1], tlo[2], tlo[3], thi[0], thi[1], thi[2], thi[3] };
        // (ring items are 8-byte aligned: 64-bit LDS reads, two slots per instruction)
        struct alignas(8) cf8 { float re, im; };
        const cf8* sp = (const cf8*)(lds + (((unsigned)sbpos & (unsigned)((MSK_RING - 1) * SLOT_B)) | (unsigned)(l * 8)));
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const cf8 t = sp[k * LPW];
            sv[k] = mk(t.re, t.im);
            tv[k] = tp[7 - k];
        }
    };
    auto fir_sum = [&](const cf* sv, const float* tv) -> cf {
        cf acc = mk(0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            acc.re += sv[k].re * tv[k];
            acc.im += sv[k].im * tv[k];
        }
        return acc;
    };

    auto fir = [&](unsigned row, int sbpos) -> cf {
        cf sv[8];
        float tv[8];
        fir_load(row, sbpos, sv, tv);
        return fir_sum(sv, tv);
    };

    // one reference iteration (:166-201), d_div of parity PAR, for the lanes exec covers
    auto body = [&](const int PAR) {
        const unsigned row = tap_row(d_mu);
        worst_imu = worst_imu > row ? worst_imu : row; // (out-of-range mu is reported)
        const cf in_interp = fir(row, sb);
        const cf sq = cmul_exact(in_interp, in_interp);                    // :171
        // :173 conj(d_dly_conj_2^2): d_dly_conj_2 is always the previous in_interp
        // (:194-195, also after a tag reset :160), so its square is the previous sq
        const cf dly_conj = cconj(prev_sq);
        const cf nlin_out = cmul_exact(sq, dly_conj);                      // :174
        float err_out = (nlin_out - d_dly_diff_1).re;                      // :178
        if (PAR) {                                                         // :179-184
            err_out = branchless_clip(err_out, 3.0f);
            d_omega += p.gain_omega * err_out;
            d_omega = d_sps + branchless_clip(d_omega - d_sps, p.limit);
            d_mu += p.gain * err_out;
        }
        if (!PAR || OSPS2) { // :186-191
            if constexpr (STG)
                stage_put(STG_MASK, stg_real, in_interp);
            else
                *(cf*)(osym0 + ob) = in_interp;
            if (AUX) {
                if (oerr0)
                    *(float*)(oerr0 + (ob >> 1)) = err_out;
                if (omu0)
                    *(float*)(omu0 + (ob >> 1)) = d_mu;
            }
            ob += 8u;
            oidx++;
        }
        d_div++;
        last_interp = in_interp; // :194-196
        prev_sq = sq;
        d_dly_diff_1 = nlin_out;
        d_mu += d_omega; // :199-201
        const float fl = floorf(d_mu);
        const int adv = (int)fl;
        iidx += adv;
        sb += adv * SLOT_B;
        d_mu = d_mu - fl;
    };

#ifdef MSK_PROF
    long long pf_t0 = __builtin_readcyclecounter(), pf_lock = 0, pf_land = 0, pf_gen = 0, pf_n[4] = {0, 0, 0, 0}, pf_fail = 0, pf_nfail = 0;
#define PF_BEGIN long long pf_a = __builtin_readcyclecounter();
#define PF_END(acc) acc += __builtin_readcyclecounter() - pf_a;
#else
#define PF_BEGIN
#define PF_END(acc)
#endif
    u64 P = 0; // parked lanes: nothing more to do before the next chunk lands (or ever)
    u64 E = 0; // lanes whose next iteration has d_div even
    // oidx < noutput was checked on the odd step before an even one and cannot have changed
    // since (osps == 1: odd iterations emit nothing), so an even step only looks at iidx
    auto ok_for = [&](const int PAR) -> bool {
        return (PAR || OSPS2) ? ((iidx < fast_lim) && (oidx < noutput)) : (iidx < fast_lim);
    };

    // general step: the lanes whose d_div has parity PAR and that are not parked
    auto step = [&](const int PAR) {
        const u64 mine = ~P & (PAR ? ~E : E);
        const u64 okM = cx.ballot(ok_for(PAR));
        u64 goM = mine & okM;
        const u64 evM = mine & ~okM;
        if (evM != 0ull) {
#ifdef MSK_PROF
            pf_n[3]++;
#endif
            int code = -1;
            if (cx.inv_ballot(evM))
                code = events(PAR);
            goM |= cx.ballot(code == EV_GO);
            P |= cx.ballot(code == EV_PARK);
            E = cx.ballot((d_div & 1) == 0);
        }
        if (goM != 0ull) {
            if (cx.inv_ballot(goM))
                body(PAR);
            E ^= goM;
        }
    };

    const u64 ALL = cx.ballot(true);

    // Bounds for the check-free pair loop.  iidx after any number of iterations is iidx0 +
    // (mu0 + the sum of the omega and gain * err terms) - (the current mu), and mu stays in
    // [0, 1]: the fraction carries over, so a run of c pairs moves iidx by less than
    // 1 + c * pair_adv, with pair_adv = 2 wmax + 3 |gain| (wmax = d_sps + |limit| bounds omega,
    // :183; |err| <= 3, :180).  The last iteration of the run starts below that.
    const float wmax = d_sps + fabsf(p.limit);
    const float pair_adv = 2.0f * wmax + 3.0f * fabsf(p.gain);
    const float pair_adv_inv = 0.9999f / pair_adv;
    // mu + omega stays positive there (so that floor is a truncation and mu - floor(mu) the
    // hardware's fract) as long as the loop filter cannot pull mu below -omega:
    const bool lock_ok = 3.0f * fabsf(p.gain) + fabsf(p.limit) + 0.01f < d_sps;
    // every lane re-arms its bound against the horizon (same formula as in events()); a lane
    // that owes the iteration after a tag reset keeps its one-iteration bound
    auto rearm = [&]() {
        const int chunk_lim = more ? (loaded_s - 8 - jump_margin + MSK_OFF - ((sb >> SLOT_SH) - iidx) + 1) : 0x7fffffff;
        int f = ninp < tag_trig ? ninp : tag_trig;
        f = f < chunk_lim ? f : chunk_lim;
        fast_lim = (tag_trig == TRIG_FORCED) ? fast_lim : f;
    };
    // (FF: a lane that has been fast-forwarded reads on elsewhere in its row: chunks go on as long as any lane wants one)
    auto need_more = [&]() -> bool {
        return FF ? cx.ballot(!done && landed * MSK_CHUNK < n + 8 - voff) != 0ull : landed < nchunks;
    };
    more = need_more();
    if (more)
        issue_chunk(landed);
    loaded_s = landed * MSK_CHUNK; // new samples [.., loaded_s) are in the rings
    P = cx.ballot(done);
    E = cx.ballot((d_div & 1) == 0);
    rearm();
    // ------------- the recurrence: every lane goes as far as its data allows -------------
    for (;;) {
        flush_syms(); // (right behind a landed chunk: the stores are done when the next one is awaited)
        // lock step: nobody parked, every lane about to run an even iteration, none at its
        // bound -> pairs of iterations run on the whole wave, exec untouched, in a loop of
        // their own (so that the values the loop carries stay in place)
        // (FF: channels whose step is over -- fast-forwarded to the end long before the others -- stand aside:
        // they run along in the lock-step runs, exec untouched, on a state that is thrown away; what they
        // were when they finished is kept here, and their stores are switched off)
        const u64 DN = FF ? cx.ballot(done) : 0ull;
        if (FF && done && !fin_saved) {
            fin_saved = true;
            fin_mu = d_mu;
            fin_omega = d_omega;
            fin_div = d_div;
            fin_interp = last_interp;
            fin_diff = d_dly_diff_1;
            fin_status = status;
            fin_worst = worst_imu;
            if constexpr (STG) { // its last symbols leave now (the owner lane staged them itself)
                if (owner) {
                    for (unsigned kk = nf; kk < (so >> SLOT_SH); kk++)
                        st8((cf*)(osym0 + obrow) + kk, stage_get(kk));
                }
                nf = so >> SLOT_SH;
            }
        }
        { PF_BEGIN
        if ((P & ~DN) == 0ull && (E | DN) == ALL && DN != ALL) {
            // How many (even, odd) pairs can EVERY lane run without looking up?  A pair moves
            // iidx by about pair_adv and emits one output (two if osps == 2), so a lane
            // with `room` items below its bound is good for room / pair_adv pairs (see pair_adv); the wave
            // takes what the slowest lane can do (at most MSK_PAIRS_MAX: one chunk's worth, the
            // next chunk is then due).  The trips run with no test at all.
            // Every iteration of a run must START below the bound.  The last one of c pairs
            // starts after c - 1 pairs and one even iteration, i.e. at most
            // 1 + (c - 1) pair_adv + wmax items further on (see pair_adv).
            // (fast_lim may be INT_MIN = "unknown" or INT_MAX = "no bound": no overflow either way)
            const float room = fast_lim > iidx ? (float)(fast_lim - iidx) - (1.001f + wmax) : -1.f;
            int can = room >= 0.f ? (int)(room * pair_adv_inv) + 1 : 0;
            can = can < MSK_PAIRS_MAX ? can : MSK_PAIRS_MAX;
            const int ocan = OSPS2 ? (noutput - oidx) / 2 : (noutput - oidx) - 1;
            can = can < ocan ? can : ocan;
            if (!lock_ok || !(d_mu >= 0.f && d_mu <= 1.f)) // (the loop below takes mu in [0, 1] for granted)
                can = 0;
            int npairs = 0; // = min over the lanes of `can`
            if ((cx.ballot(can >= 1) | DN) == ALL) { // (the usual way out near an event: one ballot)
                npairs = MSK_PAIRS_MAX;
                if ((cx.ballot(can >= MSK_PAIRS_MAX) | DN) != ALL) {
                    npairs = 1;
                    for (int bit = MSK_PAIRS_MAX / 2; bit; bit >>= 1)
                        if (npairs + bit < MSK_PAIRS_MAX && (cx.ballot(can >= npairs + bit) | DN) == ALL)
                            npairs += bit;
                }
            }
            // ---- Tag resets in line.  No plain run is possible because some lane's time_est tag
            // is about to fire (:140-164).  Instead of handing the wave to the general steps (one
            // event per pass, both iterations of every lane one after the other), pairs go on in
            // lock step with the tag tests folded in:
            //   * before the even iteration: the front tag fires there -> mu, iidx and omega are
            //     taken from the tag (d_div = 0 keeps the parity: the iteration is even anyway);
            //   * before the odd iteration: if the (next) front tag fires there, the reference
            //     runs an EVEN iteration in that place (d_div = 0) -- this channel's pair then
            //     ends after its even half (no loop filter, state as the even iteration left it)
            //     and the tag fires again, by the same test, before the even iteration of the
            //     next trip.
            // A trip moves iidx by less than pair_adv plus one tag jump (< d_sps) and one item.
            if constexpr (NQ >= 2 && !AUX && !OSPS2) {
              if (npairs == 0 && tame_tags) {
                constexpr int ROW = LPW <= 16 ? 16 : 32;
                const int chunk_lim2 = more ? (loaded_s - 8 - jump_margin + MSK_OFF - ((sb >> SLOT_SH) - iidx) + 1) : 0x7fffffff;
                const int lim2 = ninp < chunk_lim2 ? ninp : chunk_lim2; // every bound but the tag's
                const float room2 = lim2 > iidx ? (float)(lim2 - iidx) - (1.001f + wmax + d_sps) : -1.f;
                int can2 = room2 >= 0.f ? (int)(room2 * (0.9999f / (pair_adv + d_sps))) + 1 : 0;
                can2 = can2 < MSK_TAG_TRIPS ? can2 : MSK_TAG_TRIPS;
                can2 = can2 < ocan ? can2 : ocan;
                if (!lock_ok || !(d_mu >= 0.f && d_mu <= 1.f) || tag_trig == TRIG_FORCED)
                    can2 = 0;
                int ntrips = 0;
                if ((cx.ballot(can2 >= 1) | DN) == ALL) {
                    ntrips = 1;
                    while (ntrips < MSK_TAG_TRIPS && (cx.ballot(can2 >= ntrips + 1) | DN) == ALL)
                        ntrips++;
                }
                if (ntrips > 0) {
                    const bool roleO = (cx.tid() & ROW) != 0;
                    const unsigned pmask = roleO ? 0u : STG_MASK, pbase = roleO ? stg_spare : stg_real;
                    cf sqO = prev_sq, sqE = mk(0.f, 0.f), acc = last_interp;
                    float nl_prev = d_dly_diff_1.re;
                    bool last_skip = false;
                    for (int k = 0; k < ntrips; k++) {
#ifdef MSK_EMU_STATS
                        if (l == 0 && q == 0) msk_stats[5]++;
#endif
                        // the front tag fires before this even iteration (:140-164)
                        const bool fireE = (nt_rel >= iidx) && ((float)nt_rel < ((float)iidx + d_sps));
                        if (cx.ballot(fireE) != 0ull) {
                            if (fireE) {
                                const float center = nt_val;
                                if (center == center) { // not NaN (:144-147)
                                    int at = nt_rel;
                                    float m = center;
                                    if (m < 0) {
                                        m++;
                                        at--;
                                    }
                                    sb += (at - iidx) * SLOT_B;
                                    iidx = at;
                                    d_mu = m;
                                    d_div = 0;
                                    d_omega = d_sps;
                                }
                                tq_pop();
                                nt_rel = (fr_rel != TQ_NONE && fr_rel - base < ninp) ? fr_rel - base : 0x7fffffff;
                            }
                        }
                        const float m1 = d_mu + d_omega;                               // :199-201, m1 > 0
                        const float muO = cx.fract(m1);
                        const int adv1 = (int)m1;
                        const int sb1 = sb + adv1 * SLOT_B;
                        const int iidxO = iidx + adv1;
                        // ... or before the odd one: that iteration then belongs to the next trip.  (A
                        // NaN tag there is only dropped, :144-147: the odd iteration runs as it is.)
                        const bool winO = (nt_rel >= iidxO) && ((float)nt_rel < ((float)iidxO + d_sps));
                        const bool nanO = winO && (nt_val != nt_val);
                        if (cx.ballot(nanO) != 0ull) {
                            if (nanO) {
                                tq_pop();
                                nt_rel = (fr_rel != TQ_NONE && fr_rel - base < ninp) ? fr_rel - base : 0x7fffffff;
                            }
                        }
                        const bool skipO = winO && !nanO;
                        cf sv[8];
                        float tv[8];
                        fir_load((unsigned)(int)rintf((roleO ? muO : d_mu) * 128.0f), roleO ? sb1 : sb, sv, tv);
                        acc = fir_sum(sv, tv);
                        const cf sq = cmul_exact(acc, acc);                            // :171
                        cf sE, s1;
                        cx.template pair_rows<ROW>(sq, sE, s1);
                        const float nlE = sE.re * sqO.re + sE.im * sqO.im;             // :173-174, real part
                        const float nlO = s1.re * sE.re + s1.im * sE.im;
                        const float err = branchless_clip(nlO - nlE, 3.0f);            // :179-184
                        float om2 = d_omega + p.gain_omega * err;
                        om2 = d_sps + branchless_clip(om2 - d_sps, p.limit);
                        const float mu2 = muO + p.gain * err;
                        if constexpr (STG) {                                           // :186-191 (even iterations only)
                            stage_put(pmask, pbase, acc);
                        } else {
                            if (!roleO)
                                *(cf*)(osym0 + ob) = acc;
                            ob += 8u;
                        }
                        const float m2 = mu2 + om2;                                    // > 0 (lock_ok)
                        const int adv2 = (int)m2;
                        d_mu = skipO ? muO : cx.fract(m2);
                        sb = skipO ? sb1 : sb1 + adv2 * SLOT_B;
                        iidx = skipO ? iidxO : iidxO + adv2;
                        d_omega = skipO ? d_omega : om2;
                        d_div += skipO ? 1 : 2;
                        // the last two squares (for d_dly_conj_2 and the imaginary part of nlin_out)
                        const cf nE = skipO ? sqO : sE, nO = skipO ? sE : s1;
                        sqE = nE;
                        sqO = nO;
                        nl_prev = skipO ? nlE : nlO;
                        last_skip = skipO;
                    }
                    cf accE, accO;
                    cx.template pair_rows<ROW>(acc, accE, accO);
                    oidx += ntrips;
                    prev_sq = sqO;
                    last_interp = last_skip ? accE : accO;
                    d_dly_diff_1 = mk(nl_prev, sqO.im * sqE.re - sqO.re * sqE.im);
                    if (!(d_mu >= 0.f && d_mu <= 1.f))
                        status |= MSK_ST_INTERP_RANGE;
                    E = cx.ballot((d_div & 1) == 0);
                    bounds(sb >> SLOT_SH); // where the (new) front tag can fire, how far this lane can run
#ifdef MSK_PROF
                    pf_n[0] += ntrips; pf_n[1]++;
#endif
                }
              }
            }
#ifdef MSK_PROF
            if (npairs == 0) { pf_fail += __builtin_readcyclecounter() - pf_a; pf_nfail++; }
#endif
            if (npairs > 0) {
              if constexpr (NQ >= 2) {
                // Two lanes per channel share a pair: lane i (in an even row of 16, or the lower half with LPW = 32) runs the even
                // iteration, lane i + 16 (i + 32 with LPW = 32) the odd one, at the same time.  The even iteration has
                // no feedback into mu (:179), so where the odd one reads is known when the pair
                // starts; each lane does ONE interpolation (one burst of LDS reads, one sum), the
                // squares cross over by a row swap (v_permlane16/32_swap), and the loop filter is
                // computed by both lanes alike.  (With LPW <= 16 the other copies of a channel do the same once more.)
                constexpr int ROW = LPW <= 16 ? 16 : 32; // lane i and lane i ^ ROW carry the same channel
                const bool roleO = (cx.tid() & ROW) != 0;
                // (the lanes running odd iterations stage theirs in the spare row: no exec masking)
                const unsigned pmask = roleO ? 0u : STG_MASK, pbase = roleO ? stg_spare : stg_real;
                cf sqO = prev_sq, sqE = mk(0.f, 0.f), acc = last_interp;
                float nl_prev = d_dly_diff_1.re;
                const int sb_entry = sb;
                for (int k = 0; k < npairs; k++) {
#ifdef MSK_EMU_STATS
                    if (l == 0 && q == 0) msk_stats[0]++;
#endif
                    const float m1 = d_mu + d_omega;                               // :199-201, m1 > 0:
                    const float muO = cx.fract(m1);                                // m1 - floorf(m1)
                    const int sb1 = sb + (int)m1 * SLOT_B;                         // (int)floorf(m1)
                    cf sv[8];
                    float tv[8];
                    fir_load((unsigned)(int)rintf((roleO ? muO : d_mu) * 128.0f), roleO ? sb1 : sb, sv, tv);
                    acc = fir_sum(sv, tv);
                    const cf sq = cmul_exact(acc, acc);                            // :171
                    cf sE, s1;
                    cx.template pair_rows<ROW>(sq, sE, s1);                        // even lane's, odd lane's
                    const float nlE = sE.re * sqO.re + sE.im * sqO.im;             // :173-174, real part
                    const float nlO = s1.re * sE.re + s1.im * sE.im;
                    const float err = branchless_clip(nlO - nlE, 3.0f);            // :179-184
                    d_omega += p.gain_omega * err;
                    d_omega = d_sps + branchless_clip(d_omega - d_sps, p.limit);
                    const float mu2 = muO + p.gain * err;
                    if (OSPS2) {                                                   // :186-191
                        const unsigned o = ob + (roleO ? 8u : 0u);
                        *(cf*)(osym0 + o) = acc;
                        if (AUX) {
                            if (oerr0)
                                *(float*)(oerr0 + (o >> 1)) = roleO ? err : nlE - nl_prev;
                            if (omu0)
                                *(float*)(omu0 + (o >> 1)) = roleO ? mu2 : d_mu;
                        }
                        ob += 16u;
                    } else if constexpr (STG) {
                        stage_put(pmask, pbase, acc);
                    } else {
                        if (!roleO) {
                            *(cf*)(osym0 + ob) = acc;
                            if (AUX) {
                                if (oerr0)
                                    *(float*)(oerr0 + (ob >> 1)) = nlE - nl_prev;
                                if (omu0)
                                    *(float*)(omu0 + (ob >> 1)) = d_mu;
                            }
                        }
                        ob += 8u;
                    }
                    const float m2 = mu2 + d_omega;                                // > 0 (lock_ok)
                    d_mu = cx.fract(m2);
                    sb = sb1 + (int)m2 * SLOT_B;
                    sqE = sE;
                    sqO = s1;
                    nl_prev = nlO;
                }
                // the state every lane of the channel carries on with
                cf accE, accO;
                cx.template pair_rows<ROW>(acc, accE, accO);
                d_div += 2 * npairs;
                oidx += OSPS2 ? 2 * npairs : npairs;
                iidx += (sb - sb_entry) >> SLOT_SH;
                prev_sq = sqO;
                last_interp = accO;
                // :174 imaginary part of the last nlin_out: only ever read back as state
                d_dly_diff_1 = mk(nl_prev, sqO.im * sqE.re - sqO.re * sqE.im);
                if (!(d_mu >= 0.f && d_mu <= 1.f)) // (non-finite input: upstream would have thrown)
                    status |= MSK_ST_INTERP_RANGE;
              } else {
                // one lane per channel (LPW = 64): both iterations of a pair in the same lane
                const int sb_entry = sb;
                cf sqO = prev_sq, sqE = mk(0.f, 0.f), accO = last_interp;
                float nl_prev = d_dly_diff_1.re;
                for (int k = 0; k < npairs; k++) {
#ifdef MSK_EMU_STATS
                    if (l == 0) msk_stats[0]++;
#endif
                    // ---- even iteration (:166-201 with d_div even).  It has no feedback into
                    // mu (:179), so where the odd iteration will read is known now: both sets of
                    // loads go out together, the odd one's latency hides behind the even sum.
                    cf svE[8], svO[8];
                    float tvE[8], tvO[8];
                    fir_load((unsigned)(int)rintf(d_mu * 128.0f), sb, svE, tvE);
                    const float m1 = d_mu + d_omega;                               // :199-201, m1 > 0:
                    const float muO = cx.fract(m1);                                // m1 - floorf(m1)
                    const int sb1 = sb + (int)m1 * SLOT_B;                            // (int)floorf(m1)
                    fir_load((unsigned)(int)rintf(muO * 128.0f), sb1, svO, tvO);
                    const cf accE = fir_sum(svE, tvE);
                    const cf sE = cmul_exact(accE, accE);                          // :171
                    const float nlE = sE.re * sqO.re + sE.im * sqO.im;             // :173-174, real part
                    if constexpr (STG)                                             // :186-191
                        stage_put(STG_MASK, stg_real, accE);
                    else
                        *(cf*)(osym0 + ob) = accE;
                    if (AUX) {
                        if (oerr0)
                            *(float*)(oerr0 + (ob >> 1)) = nlE - nl_prev;
                        if (omu0)
                            *(float*)(omu0 + (ob >> 1)) = d_mu;
                    }
                    ob += 8u;
                    // ---- odd iteration: loop filter (:179-184)
                    const cf acc1 = fir_sum(svO, tvO);
                    const cf s1 = cmul_exact(acc1, acc1);
                    const float nlO = s1.re * sE.re + s1.im * sE.im;
                    const float err = branchless_clip(nlO - nlE, 3.0f);
                    d_omega += p.gain_omega * err;
                    d_omega = d_sps + branchless_clip(d_omega - d_sps, p.limit);
                    const float mu2 = muO + p.gain * err;
                    if (OSPS2) {
                        *(cf*)(osym0 + ob) = acc1;
                        if (AUX) {
                            if (oerr0)
                                *(float*)(oerr0 + (ob >> 1)) = err;
                            if (omu0)
                                *(float*)(omu0 + (ob >> 1)) = mu2;
                        }
                        ob += 8u;
                    }
                    const float m2 = mu2 + d_omega;                                // > 0 (lock_ok)
                    d_mu = cx.fract(m2);
                    sb = sb1 + (int)m2 * SLOT_B;
                    sqE = sE;
                    sqO = s1;
                    accO = acc1;
                    nl_prev = nlO;
                }
                // hand the state back to the per-lane variables
                d_div += 2 * npairs;
                oidx += OSPS2 ? 2 * npairs : npairs;
                iidx += (sb - sb_entry) >> SLOT_SH;
                prev_sq = sqO;
                last_interp = accO;
                // :174 imaginary part of the last nlin_out: only ever read back as state
                d_dly_diff_1 = mk(nl_prev, sqO.im * sqE.re - sqO.re * sqE.im);
                if (!(d_mu >= 0.f && d_mu <= 1.f)) // (non-finite input: upstream would have thrown)
                    status |= MSK_ST_INTERP_RANGE;
              }
#ifdef MSK_PROF
                pf_n[0] += npairs; pf_n[1]++;
#endif
            }
        }
        PF_END(pf_lock) }
        // (some lane needs attention.)  The next chunk lands as soon as no lane still reads
        // the slots it overwrites: samples [64t - 256, 64t - 192) for chunk t (a lane may step
        // back one item on a tag, :151-154) -- usually before any lane has to wait for it.
        // Lanes that do wait satisfy the condition themselves, so when everybody waits the
        // chunk does land.
        { PF_BEGIN
        if (FF && !more) { // (a fast-forward may have put a lane where the chunks had stopped coming)
            more = need_more();
            if (more)
                issue_chunk(landed);
        }
        if (more) {
            const bool clear = done || ((sb >> SLOT_SH) - MSK_OFF >= landed * MSK_CHUNK - (MSK_RING - MSK_CHUNK) + 2);
            if (cx.ballot(clear) == ALL) {
#ifdef MSK_EMU_STATS
                if (l == 0) msk_stats[4]++;
#endif
                cx.wave_sync(); // (lane model: every lane is done with the slots about to be overwritten)
                land_chunk(landed);
                cx.wave_sync(); // (lane model: the shares the other lanes wrote are in place)
                landed++;
                more = need_more();
                if (more)
                    issue_chunk(landed);
                loaded_s = landed * MSK_CHUNK;
                P = cx.ballot(done); // whoever waited goes on
                rearm();
                if (P == 0ull && E == ALL) {
                    PF_END(pf_land)
                    continue;
                }
            }
        }
        PF_END(pf_land) }
#ifdef MSK_EMU_STATS
        if (l == 0) { msk_stats[1]++; if (P != 0ull) msk_stats[2]++; if (E != ALL && E != 0ull) msk_stats[3]++; }
#endif
        { PF_BEGIN
        step(0);
        step(1);
#ifdef MSK_PROF
        pf_n[2]++;
#endif
        PF_END(pf_gen) }
        if (P == ALL && (!more || cx.ballot(done) == ALL))
            break; // all parked and nothing they could wait for: all done
    }

#ifdef MSK_PROF
    if (c == 0 && owner) {
        long long pf_t1 = __builtin_readcyclecounter();
        printf("msk prof: total %lld lock %lld (failed entries %lld: %lld) land %lld gen %lld | pairs %lld runs %lld genpasses %lld events %lld\n", pf_t1 - pf_t0, pf_lock, pf_nfail, pf_fail, pf_land, pf_gen, pf_n[0], pf_n[1], pf_n[2], pf_n[3]);
    }
#endif
    if constexpr (STG) { // what is left in the stage: fewer than FL symbols per channel, 8-byte stores
        flush_syms();
        const unsigned left = fin_saved ? 0u : (so >> SLOT_SH) - nf;
#pragma unroll
        for (unsigned j = 0; j < 2; j++) {
            const unsigned k = (unsigned)q + j * FLQ;
            if ((unsigned)q < FLQ && k < left)
                st8((cf*)(osym0 + obrow) + nf + k, stage_get(nf + k));
        }
    }
    if (!live || !owner)
        return;
    if (fin_saved) {
        d_mu = fin_mu;
        d_omega = fin_omega;
        d_div = fin_div;
        last_interp = fin_interp;
        d_dly_diff_1 = fin_diff;
        status = fin_status;
        worst_imu = fin_worst;
    }
    if (worst_imu >= (unsigned)MSK_ZERO_ROW)
        status |= MSK_ST_INTERP_RANGE;
    p.mu[c] = d_mu;
    p.omega[c] = d_omega;
    p.div[c] = d_div;
    p.dly1[c] = last_interp;
    p.dly2[c] = last_interp;
    p.diff1[c] = d_dly_diff_1;
    const unsigned long long Rn = R + (unsigned long long)base;
    p.nread[c] = Rn;
    p.produced[c] = ototal;
    p.consumed[c] = base;

    cf* cout = p.carry_out + (long)c * p.carry_cap;
    if (p.stream_mode) {
        int left = navail - base; // pending items for the next call
        int cap = p.carry_cap < MSK_CARRY_MAX ? p.carry_cap : MSK_CARRY_MAX;
        if (left + 1 > cap) {
            status |= MSK_ST_CARRY_OVERFLOW;
            left = cap - 1;
        }
        if (FF) { // (the ring may stand elsewhere in the row: from memory)
            for (int k = 0; k <= left; k++) {
                const int qi = base - 1 + k; // logical item: -1 the item before nitems_read, then the carried, then the new ones
                cout[k] = qi < pending ? cin[qi + 1] : myin[qi - pending];
            }
            p.npieces[c] = ffnp;
        } else {
            for (int k = 0; k <= left; k++)
                cout[k] = myring[((base - 1 + k - pending + MSK_OFF) & (MSK_RING - 1)) * LPW];
        }
        p.carry_len_out[c] = left;
        // tags the scheduler still holds: offset >= nitems_read
        tag_rec* cto = p.ctag_out + (long)c * p.ctag_cap;
        int w = 0;
        for (int k = 0; k < qn; k++) { // queued ones (value already narrowed to the float the loop uses)
            const tq_ent e = tq[k * TQS];
            if (e.rel < base)
                continue;
            if (w < p.ctag_cap) {
                tag_rec tg;
                tg.offset = R + (unsigned long long)e.rel;
                tg.value = (double)e.val;
                tg.key = KEY_TIME_EST;
                tg.chan = c;
                cto[w] = tg;
            } else {
                status |= MSK_ST_TAGCARRY_OVERFLOW;
            }
            w++;
        }
        for (int k = gq; k < ntot; k++) { // and the ones never queued
            const tq_ent e = ctl[k];
            if (e.rel < base)
                continue;
            if (w < p.ctag_cap) {
                tag_rec tg;
                tg.offset = R + (unsigned long long)e.rel;
                tg.value = (double)e.val;
                tg.key = KEY_TIME_EST;
                tg.chan = c;
                cto[w] = tg;
            } else {
                status |= MSK_ST_TAGCARRY_OVERFLOW;
            }
            w++;
        }
        p.ctag_n_out[c] = w < p.ctag_cap ? w : p.ctag_cap;
    } else {
        cout[0] = myring[((base - 1 - pending + MSK_OFF) & (MSK_RING - 1)) * LPW];
        p.carry_len_out[c] = 0;
        p.ctag_n_out[c] = 0;
    }
    p.status[c] = status;
}

// Tag prepass: one wave per channel merges the tags carried over from the previous call with
// this call's, keeps the time_est ones at or after nitems_read (:125-130 asks for that key
// only) and writes them as msk_ctag.  Runs before msk_body on the same stream.
struct TagPrepParams {
    int nchan;
    const tag_rec* ctag_in; const int* ctag_n_in; int ctag_cap; // carried (already time_est only)
    const tag_rec* tags; const int* tag_count; int tag_cap;     // this call's, any keys (may be null)
    const unsigned long long* nread;
    msk_ctag* ct; int* ct_n; int ct_cap;
    int* ct_nc; // (may be null) carried tags kept, in front of the new ones; -1: one of the new tags was dropped
    // (may be null) this call's time_est tags as k_mskp.h's prepass left them (row offsets): taken instead of `tags`
    const msk_ctag* ctl_new; const int* ctl_new_n; int ctl_new_cap; int ctl_new_pre; unsigned long long W;
};

template <class Ctx>
AISX_DI void tagprep_body(Ctx& cx, const TagPrepParams& p)
{
    // one wave per channel: 64 records per step, coalesced; ballot + popcount compaction
    // keeps the list order
    const int l = cx.tid() & 63;
    const int c = cx.bx() * (cx.nthreads() >> 6) + (cx.tid() >> 6);
    if (c >= p.nchan)
        return;
    const unsigned long long R = p.nread[c];
    msk_ctag* out = p.ct + (long)c * p.ct_cap;
    int w = 0; // wave-uniform
    bool wild = false;
    auto scan = [&](const tag_rec* list, int n) {
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int k = k0 + l;
            tag_rec t;
            t.offset = 0;
            t.value = 0;
            t.key = -1;
            if (k < n)
                t = list[k];
            const bool keep = (k < n) && t.key == KEY_TIME_EST && t.offset >= R;
            const float tv = (float)t.value;
            if (cx.ballot(keep && !(tv != tv) && !(tv >= -1.0f && tv <= 1.0f)) != 0ull)
                wild = true;
            const unsigned long long m = cx.ballot(keep);
            const int pos = w + aisx_popc64(m & ((1ull << l) - 1ull));
            if (keep && pos < p.ct_cap) {
                const unsigned long long d = t.offset - R;
                msk_ctag e;
                e.rel = d > 0x7ffffff0ull ? 0x7ffffff0 : (int)d;
                e.val = (float)t.value;
                out[pos] = e;
            }
            w += aisx_popc64(m);
        }
    };
    bool trunc = false;
    int nc = p.ctag_n_in[c];
    if (nc > p.ctag_cap)
        nc = p.ctag_cap;
    scan(p.ctag_in + (long)c * p.ctag_cap, nc);
    const int kept_carried = w;
    bool newdrop = false;
    if (p.ctl_new) {
        int nn = p.ctl_new_n[c];
        if (nn & MSK_CTN_TRUNC)
            trunc = true;
        if (nn & MSK_CTN_WILD)
            wild = true;
        nn &= MSK_CTN_WILD - 1;
        const msk_ctag* list = p.ctl_new + (long)c * p.ctl_new_cap + p.ctl_new_pre;
        const long long shift = (long long)(p.W - R); // row item 0 relative to nitems_read
        for (int k0 = 0; k0 < nn; k0 += 64) {
            const int k = k0 + l;
            msk_ctag e;
            e.rel = 0;
            e.val = 0.f;
            if (k < nn)
                e = list[k];
            const long long d = shift + (long long)e.rel;
            const bool keep = (k < nn) && d >= 0; // (offset >= nitems_read)
            if (cx.ballot((k < nn) && !keep) != 0ull)
                newdrop = true;
            const unsigned long long m = cx.ballot(keep);
            const int pos = w + aisx_popc64(m & ((1ull << l) - 1ull));
            if (keep && pos < p.ct_cap) {
                msk_ctag o;
                o.rel = d > 0x7ffffff0ll ? 0x7ffffff0 : (int)d;
                o.val = e.val;
                out[pos] = o;
            }
            w += aisx_popc64(m);
        }
    } else if (p.tags) {
        int nn = p.tag_count[c];
        if (nn > p.tag_cap) { // the producer (corr_est) ran out of room: the list is incomplete
            nn = p.tag_cap;
            trunc = true;
        }
        const int w0 = w;
        scan(p.tags + (long)c * p.tag_cap, nn);
        if (p.ct_nc) { // (how many of the new tags have the key: all of them must have been kept)
            int nkey = 0;
            const tag_rec* list = p.tags + (long)c * p.tag_cap;
            for (int k0 = 0; k0 < nn; k0 += 64) {
                const int k = k0 + l;
                nkey += aisx_popc64(cx.ballot(k < nn && list[k].key == KEY_TIME_EST));
            }
            newdrop = (w - w0) != nkey;
        }
    }
    if (w > p.ct_cap)
        trunc = true;
    if (l == 0) {
        p.ct_n[c] = (w < p.ct_cap ? w : p.ct_cap) | (trunc ? MSK_CTN_TRUNC : 0) | (wild ? MSK_CTN_WILD : 0);
        if (p.ct_nc)
            p.ct_nc[c] = (newdrop || trunc) ? -1 : kept_carried;
    }
}

// Bit tail (python/ais_demod.py:48-52, lib/invert_impl.cc:62-64): workgroup (seg, ch)
// turns symbols [seg * BT_SEG, +BT_SEG) of channel ch into bits.
//   b[o]   = binary_slicer(quadrature_demod: pi/2 * fast_atan2f(sym[o] * conj(sym[o-1])))
//   bit[o] = ((b[o] - b[o-1]) mod 2) ^ 1
// with sym[-1] / b[-1] the state left by the previous call.
template <class Ctx>
AISX_DI void bittail_body(Ctx& cx, const BitTailParams& p)
{
    const int ch = cx.by();
    const int t = cx.tid();
    float* at = (float*)cx.lds();
    for (int i = t; i < 257; i += BT_T)
        at[i] = p.atan_tab[i];
    cx.sync();
    const int P = p.produced[ch];
    const cf* sy = p.syms + (long)ch * p.sym_stride;
    unsigned char* ob = p.bits + (long)ch * p.bit_stride;
    const cf sym_m1 = p.prev_sym_in[ch];
    const unsigned char bit_m1 = p.prev_bit_in[ch];
    // binary_slicer(quadrature_demod) only wants the SIGN of pi/2 * fast_atan2f(y, x).  For finite
    // operands every arm of fast_atan2f gives an angle >= 0 when y >= 0 (-0 included: base_angle,
    // pi - base_angle, pi/2 -+ base_angle with base_angle <= pi/4; 0 for x = y = 0) and an angle < 0
    // when y < 0 -- except -base_angle = -0 (which the slicer takes for 1) when the quotient
    // |y| / |x| underflows to zero.  So: the sign of y decides, unless an operand is not finite or
    // a negative y is tiny against x; those (never, on symbols) go through the table.
    auto slice = [&](const cf& cur, const cf& prev) -> unsigned char {
        const cf prod = cmul_exact(cur, cconj(prev));
        const float y = prod.im, x = prod.re;
        const float ya = fabsf(y), xa = fabsf(x);
        const bool plain = (ya < 1.0e30f) && (xa < 1.0e18f) && (y >= 0.0f || ya > 1.0e-18f); // (false for NaN / inf)
        if (plain)
            return y >= 0.0f ? 1 : 0;
        const float fm = 1.57079632679489661923f * fast_atan2f_tab(y, x, at);
        return fm >= 0 ? 1 : 0;
    };
    if (cx.bx() == 0 && t == 0 && P == 0) { // nothing produced: the state carries over
        p.prev_sym_out[ch] = sym_m1;
        p.prev_bit_out[ch] = bit_m1;
    }
    const int o0 = cx.bx() * BT_SEG;
    for (int k = 0; k < BT_SEG / BT_T; k++) {
        const int o = o0 + k * BT_T + t;
        if (o >= P)
            break;
        const cf s0 = sy[o];
        const cf s1 = o >= 1 ? sy[o - 1] : sym_m1;
        const unsigned char b = slice(s0, s1);
        unsigned char bp = bit_m1;
        if (o >= 1)
            bp = slice(s1, o >= 2 ? sy[o - 2] : sym_m1);
        const unsigned char d = (unsigned char)(((unsigned)(b - bp)) % 2u);
        ob[o] = (unsigned char)((d ^ 0x01) & 0x01);
        if (o == P - 1) {
            p.prev_sym_out[ch] = s0;
            p.prev_bit_out[ch] = b;
        }
    }
}

} // namespace aisx
