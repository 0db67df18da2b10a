// k_mskp.h -- msk_timing_recovery_cc (reference: lib/msk_timing_recovery_cc_impl.cc:107-206),
// parallel in time.
//
// The reference loop is a recurrence through (d_mu, d_omega, iidx, the two delay registers).  But a
// time_est tag resets d_mu, iidx, d_div and d_omega from the tag alone (:140-164), and corr_est
// leaves its time_est tags in PAIRS one symbol apart (two preamble peaks): when tag A resets the
// loop, the iteration after it (A: d_div even, no loop filter, :179) and the next one (B) read
// samples at positions that follow from tag A only, and when tag B then resets the loop two
// iterations later the whole state -- d_mu, iidx, d_div = 0, d_omega = d_sps from tag B;
// d_dly_conj_2 = y_B; d_dly_diff_1 = y_B^2 conj(y_A^2) -- is a function of the two tags and the
// samples.  Such a pair is a RESTART POINT: the loop can be entered there without knowing
// anything that came before.  Whether the serial loop really does pass through that state is
// history (tag B is stepped over when iteration B advances by three samples, or blocked behind
// an older stale tag, :142), so it is CHECKED, bit for bit, and what does not check is re-run.
//
//   mskp_prep_body : one wave per channel.  Compacts this call's time_est tags to (row offset,
//                    float value), picks restart points about n / smax samples apart.
//   mskp_body<SPEC>: one LANE per (channel, restart point): enters at the restart point and runs
//                    the reference iteration to the next restart point; leaves its symbols in a
//                    staging row and its end state in memory.  Needs nothing from the previous
//                    call (row coordinates: item 0 = first new item of this call).
//   mskp_body<JOIN>: one lane per channel: the scheduler of the stream contract (DESIGN.md
//                    section 2) and the serial loop from the carried state.  When the front tag
//                    is a restart point's tag B about to fire and the two delay registers equal
//                    the ones that unit assumed, the unit's run IS the serial loop's: the lane
//                    takes the unit's symbol count and end state and goes on from there -- through
//                    the following units too while each one ended on the next one's assumption.
//                    Anything else (a junction that does not check, a stale tag, the end of the
//                    general_work call) it runs itself.
//   mskp_gather_body: copies the accepted units' symbols from the staging rows into the output row.
//
// Both kernels run the same engine: every lane keeps the last 64 samples of ITS stream in an
// LDS ring (slot-major: ring[slot][lane], conflict-free however far the lanes are apart), the
// rings are refilled in lock step (block L of eight samples of every lane at once, by LDS-DMA --
// `buffer_load_dwordx4 ... lds`, 16 bytes per lane straight into the ring row, no registers --
// issued MSKP_D blocks ahead and awaited with an exact s_waitcnt vmcnt(n): the symbol stores and
// the younger blocks stay in flight), and a trip of the loop runs one even and one odd iteration per
// lane with the tag tests in line, predicated, no divergence.  All arithmetic is the
// reference's float sequence, unfused: symbols are bit-identical to the serial kernel's
// (k_msk.h), which stays for osps = 2, the err / mu ports and the GNU Radio work call.
#pragma once
#include "aisx_common.h"
#include "k_msk.h"

namespace aisx {

#ifndef MSKP_R
#define MSKP_R 128                       // ring slots per lane (power of two)
#endif
#ifndef MSKP_BS
#define MSKP_BS 16                       // samples of a lane per block (a multiple of 8)
#endif
#ifndef MSKP_D
#define MSKP_D 3                         // blocks in flight (LDS-DMA) ahead of the last one readable
#endif
constexpr int MSKP_NB = MSKP_R / MSKP_BS; // blocks in the ring
constexpr int MSKP_ROWS = MSKP_R / 2;    // ring rows: row r = two consecutive samples of every lane, [row][lane] 16 bytes each
constexpr int MSKP_SLOTS = MSKP_R + 8;   // + 4 mirror rows: five rows from any row never wrap
static_assert(MSKP_BS % 8 == 0 && (MSKP_NB - MSKP_D) * MSKP_BS >= 24, "readable window of the rings");
constexpr int MSKP_TQ = 8;               // time_est tags staged in LDS per lane
constexpr int MSKP_TPRE = 64;            // room in front of a channel's new-tag list for the carried tags
constexpr int MSKP_PREP_LDS_TAGS = 2048; // tags of a channel the restart search looks at
constexpr int MSKP_GATHER_X = 4;         // workgroups per channel of the gather kernel
constexpr int MSKP_NCLS = 7;             // units are handed to the waves by length class: < 256 items, < 512, ... >= 8192
constexpr int MSKP_WALK_EVERY = 32;      // trips a lane waits at a junction at most while others still run
// The rings: four groups of 16 lanes; a group's ring is MSKP_ROWS + 4 rows of 256 bytes, row r = the
// sample pair (2 r, 2 r + 1) of each of its 16 lanes, 16 bytes per lane.  A block of eight samples of
// a group = four rows = 1 KiB, which ONE LDS-DMA instruction fills: lane 16 q + u brings pair q of lane
// u's stream (the four lanes of a stream read 64 contiguous bytes: 16 cache lines per instruction,
// not 64).  The four rows behind the ring mirror its first four: five rows from any row never wrap.
// A lane's 16-byte reads hit bank group (lane & 7) wherever its row is: no conflicts.
constexpr int MSKP_GRP_B = (MSKP_ROWS + 4) * 256;
constexpr int MSKP_LDS_RING = 4 * MSKP_GRP_B;
static_assert(MSKP_LDS_RING == MSKP_SLOTS * 64 * 8, "ring bytes");
constexpr int MSKP_LDS_TQ = MSKP_TQ * 64 * 8;
// units kernel: 16 symbols per lane staged for 64-byte stores, [group][slot pair][lane of the group] 16 bytes
constexpr int MSKP_LDS_SYM = 16 * 64 * 8;
constexpr int MSKP_LDS_BYTES = MSKP_LDS_RING + MSKP_LDS_TQ + MSK_LDS_MMSE + MSKP_LDS_SYM;

struct MskpParams {
    int nchan;
    float d_sps, gain, gain_omega, limit; // loop constants (set_sps / set_gain / set_limit, :69-96)
    // per-channel loop state and carry, as k_msk.h keeps them (the two kernels are interchangeable call by call)
    float* mu; float* omega; int* div;
    cf* dly1; cf* dly2; cf* diff1;
    unsigned long long* nread;
    const cf* in; long in_stride; int n;
    const cf* carry_in; cf* carry_out; const int* carry_len_in; int* carry_len_out; int carry_cap;
    const tag_rec* ctag_in; const int* ctag_n_in; tag_rec* ctag_out; int* ctag_n_out; int ctag_cap;
    // this call's time_est tags, compacted by the prepass: entry k of channel c at ctl[c * ctl_cap + MSKP_TPRE + k]
    msk_ctag* ctl; const int* ctl_n; int ctl_cap;
    // restart points and what the units made of them
    int smax; const int* nrst; const mskp_rst* rst; mskp_res* res;
    cf* stage; long stage_stride;
    // output
    cf* syms; long out_stride; int out_cap;
    mskp_piece* pieces; int* npieces;
    int* produced; int* consumed; int* status;
    const float* mmse;          // [129][8]
    unsigned long long W;       // absolute offset of row item 0 (items handed to the block before this call)
    int look;                   // samples a trip may read beyond in[iidx]
    int padv;                   // items a pair of iterations moves iidx at most
    float padv_inv;             // a run of c pairs moves iidx by less than 1 + c / padv_inv
    int jw;                     // channels per wave of the join kernel (1..64: fewer lanes, fewer events per trip)
    // units by length class (null: lane u of wave w takes unit (w * 64 + u) % smax of channel (w * 64 + u) / smax)
    const int* ucount; const int* ulist; long ucap;
    int tail;                   // units stop this many items before the end of the row
    int max_noutput;            // set_max_noutput_items(): output items a general_work call is offered at most (0: what fits)
};

struct MskpPrepParams {
    int nchan;
    const tag_rec* tags; const int* tag_count; int tag_cap; // this call's tags, any keys (may be null)
    unsigned long long W;
    int n;
    float d_sps, gain, limit;
    msk_ctag* ctl; int* ctl_n; int ctl_cap;
    int smax; int* nrst; mskp_rst* rst;
    long stage_stride;
    int tail;
    int min_gap;    // restart points at least this many items apart
    int max_span;   // no unit runs from a restart point more than this many items before the next (the join's serial loop is faster on a long stretch)
    int* ucount; int* ulist; long ucap; // units by length class (may be null)
};

struct MskpGatherParams {
    int nchan;
    const mskp_piece* pieces; const int* npieces;
    const cf* stage; long stage_stride;
    cf* syms; long out_stride;
};

// ---- what the host derives from the loop constants
// lanes of the speculative kernel test `offset < iidx + d_sps` (:142) with iidx counted from the
// start of the row, the reference from nitems_read: the float sum must round to the same side of
// every integer for both, i.e. d_sps is an integer or well away from one (iidx < 2^18: ulp 2^-6)
AISX_HD bool mskp_geometry_ok(float d_sps, float gain, float limit, int max_items)
{
    const float fr = d_sps - floorf(d_sps);
    const bool coord_ok = fr == 0.f || (fr > 0.04f && fr < 0.96f);
    const bool lock_ok = 3.0f * fabsf(gain) + fabsf(limit) + 0.01f < d_sps; // mu + omega stays positive
    const float m = 2.0f * (d_sps - fabsf(limit)) - 3.0f * fabsf(gain);
    return coord_ok && lock_ok && m > 1.0f && d_sps <= 12.f && max_items <= (1 << 18) - 1024;
}
AISX_HD int mskp_look(float d_sps, float limit) { return (int)ceilf(d_sps) + (int)floorf(1.f + d_sps + fabsf(limit)) + 8; }
AISX_HD int mskp_padv(float d_sps, float gain, float limit) { return (int)ceilf(2.f * (d_sps + fabsf(limit)) + 3.f * fabsf(gain)) + 1; }
AISX_HD float mskp_padv_inv(float d_sps, float gain, float limit) { return 0.9999f / (2.f * (d_sps + fabsf(limit)) + 3.f * fabsf(gain)); }
AISX_HD int mskp_tail(float d_sps) { return 64 + (int)ceilf(3.f * d_sps) + (int)ceilf(d_sps); }
// staging slots per channel (the prepass drops restart points that would not fit)
AISX_HD long mskp_stage_stride(int max_items, float d_sps, float gain, float limit)
{
    const float m = 2.0f * (d_sps - fabsf(limit)) - 3.0f * fabsf(gain);
    return (long)((float)(max_items + 16 * MSKP_SMAX) / m) + 2 * MSKP_PREP_LDS_TAGS + 17 * MSKP_SMAX + 64;
}

AISX_HD bool mskp_tame(float v) { return v >= -1.0f && v <= 1.0f; } // (false for NaN)
// ---------------------------------------------------------------------------------------------
// Prepass: one wave per channel.
// ---------------------------------------------------------------------------------------------
template <class Ctx>
AISX_DI void mskp_prep_body(Ctx& cx, const MskpPrepParams& p)
{
    const int l = cx.tid() & 63;
    const int c = cx.bx() * (cx.nthreads() >> 6) + (cx.tid() >> 6);
    msk_ctag* const lt = (msk_ctag*)cx.lds() + (cx.tid() >> 6) * MSKP_PREP_LDS_TAGS;
    if (c >= p.nchan)
        return;
    msk_ctag* out = p.ctl + (long)c * p.ctl_cap + MSKP_TPRE;
    const int room = p.ctl_cap - MSKP_TPRE;
    int w = 0; // wave-uniform
    bool wild = false, trunc = false;
    if (p.tags) {
        int nn = p.tag_count[c];
        if (nn > p.tag_cap) { // the producer (corr_est) ran out of room: the list is incomplete
            nn = p.tag_cap;
            trunc = true;
        }
        const tag_rec* list = p.tags + (long)c * p.tag_cap;
        for (int k0 = 0; k0 < nn; k0 += 64) {
            const int k = k0 + l;
            tag_rec t;
            t.offset = 0;
            t.value = 0;
            t.key = -1;
            if (k < nn)
                t = list[k];
            const bool keep = (k < nn) && t.key == KEY_TIME_EST; // (:125-130 asks for that key only)
            const float tv = (float)t.value;
            if (cx.ballot(keep && !mskp_tame(tv)) != 0ull)
                wild = true;
            const unsigned long long m = cx.ballot(keep);
            const int pos = w + aisx_popc64(m & ((1ull << l) - 1ull));
            if (keep && pos < room) {
                const long long d = (long long)(t.offset - p.W);
                msk_ctag e;
                e.rel = d > 0x3ffffff0ll ? 0x3ffffff0 : (d < -0x3ffffff0ll ? -0x3ffffff0 : (int)d);
                e.val = tv;
                out[pos] = e;
                if (pos < MSKP_PREP_LDS_TAGS)
                    lt[pos] = e;
            }
            w += aisx_popc64(m);
        }
    }
    if (w > room) {
        w = room;
        trunc = true;
    }
    cx.wave_lds_sync();
    if (l != 0)
        return;
    p.ctl_n[c] = w | (trunc ? MSK_CTN_TRUNC : 0) | (wild ? MSK_CTN_WILD : 0);
    // ---- restart points: pairs (k, k + 1) the loop is expected to pass as A, B, reset (see the header)
    mskp_rst* rs = p.rst + (long)c * MSKP_SMAX;
    int K = 0;
    const int nn = w < MSKP_PREP_LDS_TAGS ? w : MSKP_PREP_LDS_TAGS;
    const int lim = p.n - p.tail; // units stop there
    const float d_sps = p.d_sps;
    // symbols a unit can emit: every pair moves iidx by at least m items, a tag adds at most two
    const float m = 2.0f * (d_sps - fabsf(p.limit)) - 3.0f * fabsf(p.gain);
    auto pair_ok = [&](int k) -> bool {
        const msk_ctag A = lt[k], B = lt[k + 1];
        if (!mskp_tame(A.val) || !mskp_tame(B.val))
            return false;
        if (A.rel < 8 || B.rel >= lim - 16 || B.rel <= A.rel)
            return false;
        int iA = A.rel;
        float mA = A.val;
        if (mA < 0) { // :150-153
            mA++;
            iA--;
        }
        const float m1 = mA + d_sps; // :199-201 after iteration A (d_omega = d_sps, :155)
        const int iB = iA + (int)floorf(m1);
        const float muB = m1 - floorf(m1);
        if ((B.rel >= iB) && ((float)B.rel < (float)iB + d_sps)) // tag B would fire one iteration early
            return false;
        const int iC = iB + (int)floorf(muB + d_sps); // where iteration C starts if the loop filter does not move it
        return B.rel >= iC - 1 && (float)B.rel < (float)(iC + 1) + d_sps;
    };
    if (p.smax > 0 && m > 0.5f && lim > 64) {
        int last = -0x40000000;
        int klast = -1; // the last pair of all: the stretch from there to the end of the row is serial
        for (int k = nn - 2; k >= 0 && klast < 0; k--)
            if (pair_ok(k))
                klast = k;
        for (int k = 0; k + 1 < nn && K < p.smax && K < MSKP_SMAX; k++) {
            const int relB = lt[k + 1].rel;
            const bool is_last = k == klast;
            // (the first pair is taken as it comes: the serial stretch in front of it is short)
            if (K > 0 && relB < last + (is_last ? 64 : p.min_gap))
                continue;
            if (!is_last && K + 1 >= p.smax && klast > k)
                continue; // keep the last slot for the last pair
            if (!pair_ok(k))
                continue;
            rs[K].jA = k;
            rs[K].relA = lt[k].rel;
            rs[K].relB = relB;
            last = relB;
            K++;
            k++; // (tag B is not some other pair's tag A)
        }
        // staging slots: unit k emits at most span / m + 2 per tag + a few symbols
        long q = 0;
        int Kfit = 0;
        for (int k = 0; k < K; k++) {
            const int end = k + 1 < K ? rs[k + 1].relB : lim;
            const int jend = k + 1 < K ? rs[k + 1].jA + 1 : nn;
            const int cap = ((int)((float)(end - rs[k].relB + 16) / m) + 2 * (jend - rs[k].jA) + 16 + 7) & ~7; // (64-byte stores)
            if (q + cap > p.stage_stride)
                break;
            rs[k].q0 = (int)q;
            rs[k].cap = (end - rs[k].relB) > p.max_span ? 0 : cap;
            q += rs[k].cap;
            Kfit = k + 1;
        }
        K = Kfit;
    }
    p.nrst[c] = K;
    if (p.ucount) {
        // a wave of the units kernel lives as long as its longest unit: units of a kind go together
        // (one atomic per class and channel, not per unit: 4096 channels hammer seven counters)
        int ncls[MSKP_NCLS], bcls[MSKP_NCLS];
        for (int i = 0; i < MSKP_NCLS; i++)
            ncls[i] = 0;
        auto cls_of = [&](int k) -> int {
            const int span = (k + 1 < K ? rs[k + 1].relB : lim) - rs[k].relB;
            int cls = 0;
            while (cls < MSKP_NCLS - 1 && span >= (256 << cls))
                cls++;
            return cls;
        };
        for (int k = 0; k < K; k++)
            if (rs[k].cap != 0)
                ncls[cls_of(k)]++;
        for (int i = 0; i < MSKP_NCLS; i++)
            bcls[i] = ncls[i] ? cx.atomic_add_i32(p.ucount + i, ncls[i]) : 0;
        for (int k = 0; k < K; k++) {
            if (rs[k].cap == 0)
                continue;
            const int cls = cls_of(k);
            p.ulist[(long)cls * p.ucap + bcls[cls]++] = (c << 6) | k;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The engine.
// ---------------------------------------------------------------------------------------------
template <class Ctx, bool JOIN>
AISX_DI void mskp_body(Ctx& cx, const MskpParams& p)
{
    typedef unsigned long long u64;
    const int lane = cx.tid() & 63;
    const int wv = cx.bx() * (cx.nthreads() >> 6) + (cx.tid() >> 6);
    char* const lds = cx.lds() + (cx.tid() >> 6) * MSKP_LDS_BYTES; // (one wave per workgroup in the product)
    msk_ctag* const tq = (msk_ctag*)(lds + MSKP_LDS_RING) + lane;          // entry e: tq[e * 64]
    float* const mm = (float*)(lds + MSKP_LDS_RING + MSKP_LDS_TQ);         // [130][MSK_TAPS_PITCH]
    char* const sbuf = lds + MSKP_LDS_RING + MSKP_LDS_TQ + MSK_LDS_MMSE;   // units: the symbol stage
    for (int i = lane; i < 129 * 8; i += 64)
        mm[(i >> 3) * MSK_TAPS_PITCH + (i & 7)] = p.mmse[i];
    if (lane < MSK_TAPS_PITCH)
        mm[MSK_ZERO_ROW * MSK_TAPS_PITCH + lane] = 0.f;
    cx.wave_lds_sync();

    // ---- which unit this lane is
    int c, k = 0;
    if (JOIN) {
        c = lane < p.jw ? wv * p.jw + lane : p.nchan;
    } else if (p.ucount) {
        // classes of long units first: wave wv is one of the ceil(count / 64) waves of its class
        int wb = 0;
        c = p.nchan;
        for (int cls = MSKP_NCLS - 1; cls >= 0; cls--) {
            const int cn = p.ucount[cls];
            const int nw = (cn + 63) >> 6;
            if (wv >= wb && wv < wb + nw) {
                const int idx = (wv - wb) * 64 + lane;
                if (idx < cn) {
                    const int id = p.ulist[(long)cls * p.ucap + idx];
                    c = id >> 6;
                    k = id & 63;
                }
            }
            wb += nw;
        }
    } else {
        const int u = wv * 64 + lane;
        c = u / p.smax;
        k = u - c * p.smax;
    }
    bool valid = c < p.nchan;
    const int cc = valid ? c : 0;
    const int K = p.nrst[cc];
    if (!JOIN && k >= K)
        valid = false;
    if (cx.ballot(valid) == 0ull)
        return;

    const float d_sps = p.d_sps;
    const int n = p.n;
    const cf* const myin = p.in + (long)cc * p.in_stride;
    const cf* const cin = p.carry_in + (long)cc * p.carry_cap;
    int pending = 0;
    if (JOIN) {
        pending = p.carry_len_in[cc];
        if (pending > MSK_CARRY_MAX)
            pending = MSK_CARRY_MAX;
    }
    const mskp_rst* const rs = p.rst + (long)cc * MSKP_SMAX;
    mskp_res* const rr = p.res + (long)cc * MSKP_SMAX;
    int status = 0;

    // ---- the tag list of this lane: (JOIN: the carried tags in front of) the new ones
    int nc = 0;
    int nn = p.ctl_n[cc];
    if (nn & MSK_CTN_TRUNC)
        status |= JOIN ? MSK_ST_TAGS_TRUNCATED : 0;
    nn &= MSK_CTN_WILD - 1;
    msk_ctag* ctl = p.ctl + (long)cc * p.ctl_cap + MSKP_TPRE;
    const u64 R = p.nread[cc];
    if (JOIN && valid) {
        // tags the scheduler still held (offset >= nitems_read), in front of the new ones
        int ncin = p.ctag_n_in[cc];
        if (ncin > p.ctag_cap)
            ncin = p.ctag_cap;
        if (ncin > MSKP_TPRE)
            ncin = MSKP_TPRE;
        const tag_rec* ci = p.ctag_in + (long)cc * p.ctag_cap;
        for (int j = ncin - 1; j >= 0; j--) {
            const tag_rec t = ci[j];
            if (t.key != KEY_TIME_EST || t.offset < R)
                continue;
            const long long d = (long long)(t.offset - p.W);
            msk_ctag e;
            e.rel = d > 0x3ffffff0ll ? 0x3ffffff0 : (d < -0x3ffffff0ll ? -0x3ffffff0 : (int)d);
            e.val = (float)t.value;
            nc++;
            ctl[-nc] = e;
        }
        ctl -= nc;
    }
    const int ntot = nc + nn;

    // ---- loop state
    int a = 0;          // iidx as a row offset
    float mu = 0.f, om = d_sps;
    int dv = 0;
    cf y = mk(0.f, 0.f), nl = mk(0.f, 0.f);
    int cur = 0, cnt = 0;
    bool running = valid;
    unsigned worst_row = 0;
    // JOIN: the general_work call in progress
    int base_row = 0, ninp_row = 0x3fffffff, nout_tot = 0x3fffffff;
    bool fin = false;       // the stream contract's step is over for this channel
    bool atj = false;       // stopped where restart point `cand`'s tag B is about to fire
    int cand = 0, cand_j = 0x7fffffff;
    int np = 0;             // pieces written
    // SPEC: where the unit ends
    bool warm = !JOIN;
    int stop_j = 0x7fffffff, lim_a = 0, q0 = 0, cap = 0x3fffffff, kind = MSKP_KIND_NONE;
    cf ay = mk(0.f, 0.f), anl = mk(0.f, 0.f);
    int relB = 0;

    // ---- tag queue
    int qhi = 0;                 // entries [.., qhi) of the list are staged (slot = index & (MSKP_TQ - 1))
    bool hasq = false, needq = false;
    int t_rel = 0;
    float t_val = 0.f;
    auto tq_front = [&]() {
        hasq = cur < qhi;
        needq = !hasq && cur < ntot;
        if (hasq) {
            const msk_ctag e = tq[(cur & (MSKP_TQ - 1)) * 64];
            t_rel = e.rel;
            t_val = e.val;
        }
    };
    auto tq_load = [&]() { // (rare) stage entries [cur, cur + MSKP_TQ) of the list
        qhi = cur + MSKP_TQ < ntot ? cur + MSKP_TQ : ntot;
        for (int j = cur; j < qhi; j++)
            tq[(j & (MSKP_TQ - 1)) * 64] = ctl[j];
        tq_front();
    };

    // ---- the rings: progress = row offset - org; block B = progress [8B, 8B + 8)
    int org = 0;
    int L = 0; // blocks readable so far (wave-uniform): progress [8 (L + MSKP_D - MSKP_NB), 8 L); MSKP_D more are in flight
    auto fetch = [&](int r) -> cf {
        cf v = mk(0.f, 0.f);
        if (r >= 0) {
            if (r < n)
                v = myin[r];
        } else if (r >= -(pending + 1)) {
            v = cin[r + pending + 1];
        }
        return v;
    };
    // block B of every lane -> rows (4 B .. 4 B + 3) & (MSKP_ROWS - 1) of its group's ring.  Lanes whose
    // eight items lie inside the row get them by LDS-DMA (from the four lanes that serve their stream);
    // at the ends of the row (JOIN: the carried items in front of it, zeros behind it) a lane fetches
    // its own through registers.
    // (units by class: the lanes of a wave come from anywhere; the host has made sure that all rows lie within 4 GiB)
    const bool anyrow = !JOIN && p.ucount != nullptr;
    const int c_lo = anyrow ? 0 : (JOIN ? wv * p.jw : (wv * 64) / p.smax); // first channel of this wave
    const long rows = anyrow ? p.nchan - 1 : (JOIN ? p.jw - 1 : 63 / p.smax + 1);
    const typename Ctx::Buf inbuf = cx.make_buf(p.in + (long)(c_lo < p.nchan ? c_lo : 0) * p.in_stride, (unsigned)((rows * p.in_stride + n) * 8));
    const unsigned lane_off = (unsigned)((long)(cc - (c_lo < p.nchan ? c_lo : 0)) * p.in_stride * 8);
    char* const ring_b = lds;
    const int ngrp = JOIN ? (p.jw + 15) >> 4 : 4; // groups of 16 lanes that carry streams
    char* const my_ring = ring_b + (lane >> 4) * MSKP_GRP_B + (lane & 15) * 16; // row r of this lane: my_ring + r * 256
    // what this lane knows of the four streams it serves (lane 16 j + (lane & 15), j = 0..3)
    int s_org[4];
    unsigned s_base[4];
    bool s_valid[4];
    bool org_dirty = true;
    auto refresh_streams = [&]() {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int src = 16 * j + (lane & 15);
            s_org[j] = cx.shfl_i32(org, src);
            s_base[j] = (unsigned)cx.shfl_i32((int)(lane_off + (unsigned)org * 8u), src);
            s_valid[j] = cx.shfl_i32(valid ? 1 : 0, src) != 0;
        }
        org_dirty = false;
    };
    auto issue_block = [&](int B) {
        if (cx.ballot(org_dirty) != 0ull)
            refresh_streams();
        const int q = lane >> 4;
        // (a lane that has stopped needs nothing more: a finished unit or channel would otherwise be
        // served past the end of its row, item by item through the slow path, for as long as the wave lives)
        const u64 runm = cx.ballot(running);
#pragma unroll
        for (int h = 0; h < MSKP_BS / 8; h++) { // eight samples of every lane at a time
            const int p0 = MSKP_BS * B + 8 * h; // progress of the first of them
            const int row0 = (p0 >> 1) & (MSKP_ROWS - 1);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (j >= ngrp)
                    continue;
                const int r0 = s_org[j] + p0;
                const bool fast = s_valid[j] && ((runm >> (16 * j + (lane & 15))) & 1ull) != 0ull && r0 >= 0 && r0 + 8 <= n;
                if (!JOIN) {
                    // (units lie inside their rows; what a stopped or absent lane's stream would bring is
                    // replaced by an offset the buffer rejects: zeros, read by nobody -- no test, no branch)
                    const unsigned off = fast ? s_base[j] + (unsigned)(p0 + 2 * q) * 8u : 0xfffffff0u;
                    cx.dma16(inbuf, off, cx.lds_addr(ring_b + j * MSKP_GRP_B + row0 * 256));
                    if (row0 == 0)
                        cx.dma16(inbuf, off, cx.lds_addr(ring_b + j * MSKP_GRP_B + MSKP_ROWS * 256));
                } else if (cx.ballot(fast) != 0ull) {
                    if (fast) {
                        const unsigned off = s_base[j] + (unsigned)(p0 + 2 * q) * 8u;
                        cx.dma16(inbuf, off, cx.lds_addr(ring_b + j * MSKP_GRP_B + row0 * 256));
                        if (row0 == 0)
                            cx.dma16(inbuf, off, cx.lds_addr(ring_b + j * MSKP_GRP_B + MSKP_ROWS * 256));
                    }
                }
            }
            const int r0 = org + p0;
            const bool slow = JOIN && valid && running && !(r0 >= 0 && r0 + 8 <= n);
            if (JOIN && cx.ballot(slow) != 0ull) {
                if (slow) {
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const cf v = fetch(r0 + j);
                        st8((cf*)(my_ring + (row0 + (j >> 1)) * 256 + (j & 1) * 8), v);
                        if (row0 == 0)
                            st8((cf*)(my_ring + (MSKP_ROWS + (j >> 1)) * 256 + (j & 1) * 8), v);
                    }
                }
            }
        }
    };
    // (blocks up to L + MSKP_D - 1 have been issued)
    auto jump_to = [&](int new_a) {
        org = ((new_a - 1) & ~7) - MSKP_BS * (L + MSKP_D);
        org_dirty = true;
    };

    // mmse_fir_interpolator_cc::interpolate(&in[iidx], mu) (:170); an imu outside [0, 128] (upstream
    // throws std::runtime_error) reads the all-zero row
    auto tap_row = [&](float m) -> unsigned {
        const unsigned imu = (unsigned)(int)rintf(m * 128.0f);
        return imu < (unsigned)MSK_ZERO_ROW ? imu : (unsigned)MSK_ZERO_ROW;
    };
    auto fir = [&](int pa, unsigned row) -> cf {
        typedef float tap4 __attribute__((vector_size(16)));
        const tap4* tp4 = (const tap4*)((const char*)mm + row * (unsigned)(MSK_TAPS_PITCH * 4));
        const tap4 tlo = tp4[0], thi = tp4[1];
        const float tp[8] = { tlo[0], tlo[1], tlo[2], tlo[3], thi[0], thi[1], thi[2], thi[3] };
        // in[iidx + 2 m] sits m rows behind in[iidx], in[iidx + 2 m + 1] m rows behind in[iidx + 1]: two
        // base addresses, eight 8-byte reads at constant offsets (the mirror rows make the fifth row safe)
        const char* a0 = my_ring + ((pa >> 1) & (MSKP_ROWS - 1)) * 256 + (pa & 1) * 8;
        const char* a1 = a0 + ((pa & 1) ? 248 : 8);
        cf acc = mk(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const cf sv = ld8((const cf*)(((j & 1) ? a1 : a0) + (j >> 1) * 256));
            acc.re += sv.re * tp[7 - j];
            acc.im += sv.im * tp[7 - j];
        }
        return acc;
    };
    // (int)floor(d_mu) of :200; non-finite or absurd values (upstream has thrown long before) move nothing
    auto adv_of = [&](float f) -> int {
        if (!(f >= -64.f && f <= 64.f)) {
            status |= MSK_ST_INTERP_RANGE;
            return 0;
        }
        return (int)f;
    };

    // ---- JOIN: the scheduler (stream contract)
    auto seek_tags = [&]() { // get_tags_in_range(nitems_read, ...) (:125-130): from the first tag at or behind nitems_read
        while (cur > 0 && ctl[cur - 1].rel >= base_row)
            cur--;
        while (cur < ntot && ctl[cur].rel < base_row)
            cur++;
        while (cand < K && rs[cand].jA + 1 + nc < cur)
            cand++;
        cand_j = cand < K ? rs[cand].jA + 1 + nc : 0x7fffffff;
        tq_load();
    };
    int ototal = 0;
    auto setup_round = [&]() {
        const int ninput = (n - base_row) - 1; // one look-ahead item is kept out of sight
        int noutput = 0;
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
        const int ninp = (int)(ninput - 3.0 * d_sps); // :119
        if (ninp <= 0 || noutput <= 0) {
            fin = true;
            running = false;
            return;
        }
        ninp_row = base_row + ninp;
        nout_tot = ototal + noutput;
        seek_tags();
    };
    auto end_call = [&]() { // this general_work() call is over (:138, :204)
        const bool progress = a > base_row;
        base_row = a; // consume_each(iidx)
        ototal = cnt;
        if (!progress) { // (a call that consumed nothing ends the step, see k_msk.h)
            fin = true;
            running = false;
            return;
        }
        setup_round();
    };

    // ---- start
    if (JOIN) {
        mu = p.mu[cc];
        om = p.omega[cc];
        dv = p.div[cc];
        y = p.dly2[cc];
        nl = p.diff1[cc];
        base_row = -pending;
        a = base_row;
        if (valid)
            setup_round();
    } else if (valid) {
        const mskp_rst me = rs[k];
        a = me.relA;
        cur = me.jA;
        relB = me.relB;
        q0 = me.q0;
        cap = me.cap;
        if (cap == 0)
            running = false; // (left to the join)
        lim_a = n - p.tail;
        if (k + 1 < K)
            stop_j = rs[k + 1].jA + 1;
        tq_load();
    }
    org = (a - 1) & ~7;
    for (int d = 0; d < MSKP_D; d++)
        issue_block(d);

    const int look = p.look;
    const int cd = (int)ceilf(d_sps);

    // one trip: an even and an odd iteration for every lane that is ready for them
    auto trip = [&]() {
        if (cx.ballot(running && needq) != 0ull) {
            if (running && needq)
                tq_load();
        }
        if (!JOIN) {
            // behind the two warm-up iterations A and B: the state at tag B, from the tags and the samples alone
            const bool patch = running && warm && dv == 2;
            if (cx.ballot(patch) != 0ull) {
                if (patch) {
                    warm = false;
                    if (cur == rs[k].jA + 1) {
                        ay = y;
                        anl = nl;
                        a = relB;
                        mu = 0.f;
                        om = d_sps;
                    } else {
                        running = false; // (the pair did not behave as a pair: nobody will ask for this unit)
                    }
                }
            }
        }
        {
            // (a lane whose samples have left the ring -- an absurd tag or mu moved it backwards --
            // has them fetched again)
            const bool lost = running && ((a - org) - 1 < MSKP_BS * (L + MSKP_D - MSKP_NB));
            if (cx.ballot(lost) != 0ull) {
                if (lost)
                    jump_to(a);
            }
        }
        const int pa = a - org;
        bool go = running && (pa + look <= MSKP_BS * L) && (pa - 1 >= MSKP_BS * (L + MSKP_D - MSKP_NB));
        // ---- top of the first iteration (:138)
        if (JOIN) {
            const bool over = go && !(cnt < nout_tot && a < ninp_row);
            if (cx.ballot(over) != 0ull) {
                if (over)
                    end_call();
            }
            go = go && !over;
        }
        const bool vis = !JOIN || t_rel < ninp_row; // (a tag at or behind nitems_read + ninp is not in this call's range)
        const bool has = hasq && vis;
        // a time_est tag lands in [iidx, iidx + d_sps) (:140-142)
        bool fire = go && has && (t_rel >= a) && ((float)(t_rel - base_row) < (float)(a - base_row) + d_sps);
        if (JOIN) {
            const bool stopj = fire && cur == cand_j;
            if (cx.ballot(stopj) != 0ull) {
                if (stopj) {
                    atj = true;
                    running = false;
                }
            }
            go = go && !stopj;
            fire = fire && !stopj;
        } else {
            // the unit ends: at the next restart point, or where speculation has to stop (a stale tag
            // blocks every later one, :142; a tag value the table cannot take; the end of the row)
            const bool s1 = fire && !warm && cur == stop_j;
            const bool s2 = go && !warm && !s1 &&
                            ((has && t_rel < a) || (fire && !mskp_tame(t_val)) || a >= lim_a || cnt >= cap - 1);
            if (cx.ballot(s1 || s2) != 0ull) {
                if (s1 || s2) {
                    kind = s1 ? MSKP_KIND_NEXT : MSKP_KIND_HANDOFF;
                    running = false;
                }
            }
            go = go && !(s1 || s2);
            fire = fire && !(s1 || s2);
        }
        const bool nan1 = t_val != t_val;
        const bool f1 = fire && !nan1;
        const bool runE = go && (((dv & 1) == 0) || f1);
        if (f1) { // :148-156
            mu = t_val;
            a = t_rel;
            if (mu < 0) {
                mu++;
                a--;
            }
            dv = 0;
            om = d_sps;
        }
        if (cx.ballot(fire) != 0ull) {
            if (fire) { // tags.erase(tags.begin()) (:145, :162)
                cur++;
                tq_front();
            }
        }
        // ---- even iteration (:166-201 with d_div even)
        {
            const unsigned row = tap_row(mu);
            const cf yi = fir(a - org, row);
            if (runE) {
                worst_row = worst_row > row ? worst_row : row;
                const cf sq = cmul_exact(yi, yi);                                // :171
                const cf nlin = cmul_exact(sq, cconj(cmul_exact(y, y)));         // :173-174
                if (JOIN || !warm) { // :187 (into the stage: flush_syms)
                    st8((cf*)(sbuf + (lane >> 4) * 2048 + ((cnt & 15) >> 1) * 256 + (lane & 15) * 16 + (cnt & 1) * 8), yi);
                    cnt++;
                }
                dv++;
                y = yi; // :194-196
                nl = nlin;
                mu += om; // :199-201
                const float fl = floorf(mu);
                a += adv_of(fl);
                mu = mu - fl;
            }
        }
        // ---- top of the second iteration (a lane whose next tag is not staged yet waits for the next trip)
        bool go2 = go && ((dv & 1) != 0) && !needq;
        {
            const int pa2 = a - org;
            go2 = go2 && (pa2 + 9 <= MSKP_BS * L) && (pa2 >= MSKP_BS * (L + MSKP_D - MSKP_NB));
        }
        if (JOIN)
            go2 = go2 && !(runE && !(cnt < nout_tot && a < ninp_row)); // (the call ends here: next trip)
        {
            const bool vis2 = !JOIN || t_rel < ninp_row;
            const bool fire2 = go2 && runE && hasq && vis2 && (t_rel >= a) &&
                               ((float)(t_rel - base_row) < (float)(a - base_row) + d_sps);
            const bool nan2 = t_val != t_val;
            if (cx.ballot(fire2 && nan2) != 0ull) {
                if (fire2 && nan2) { // :144-147: dropped, the iteration runs as it is
                    cur++;
                    tq_front();
                }
            }
            if (!JOIN && warm && fire2 && !nan2)
                running = false; // (tag B fires one iteration early: not a restart point)
            go2 = go2 && !(fire2 && !nan2); // a reset: that iteration is an even one, next trip
        }
        // ---- odd iteration (:166-201 with d_div odd: the loop filter, :179-184)
        {
            const unsigned row = tap_row(mu);
            const cf yi = fir(a - org, row);
            if (go2) {
                worst_row = worst_row > row ? worst_row : row;
                const cf sq = cmul_exact(yi, yi);
                const cf nlin = cmul_exact(sq, cconj(cmul_exact(y, y)));
                float err = (nlin - nl).re;                                      // :178
                err = branchless_clip(err, 3.0f);
                om += p.gain_omega * err;
                om = d_sps + branchless_clip(om - d_sps, p.limit);
                mu += p.gain * err;
                dv++;
                y = yi;
                nl = nlin;
                mu += om;
                const float fl = floorf(mu);
                a += adv_of(fl);
                mu = mu - fl;
            }
        }
    };

    // The same trip when nothing unusual is due in any running lane: every lane at an even iteration
    // with its samples in the ring, no call or unit about to end, no junction, no stale / NaN / absurd
    // tag, the tag behind the front one staged.  A tag reset is taken in line (:140-164), both
    // interpolations are fetched together (where the odd iteration reads follows from the even one's
    // state, :199-201 -- it has no feedback into mu, :179), no branch inside.  Returns false, having
    // done nothing, when some lane needs trip().
    auto fast_trip = [&]() -> bool {
        const int pa = a - org;
        const bool ready = (pa + look <= MSKP_BS * L) && (pa - 1 >= MSKP_BS * (L + MSKP_D - MSKP_NB));
        const bool vis = !JOIN || t_rel < ninp_row;
        const bool has = hasq && vis;
        const bool fire = has && (t_rel >= a) && ((float)(t_rel - base_row) < (float)(a - base_row) + d_sps);
        bool rare = ((dv & 1) != 0) || needq || (fire && (!mskp_tame(t_val) || (cur + 1 >= qhi && cur + 1 < ntot)));
        if (JOIN)
            rare = rare || !(cnt + 1 < nout_tot && a + look < ninp_row) || (fire && cur == cand_j);
        else
            rare = rare || warm || a >= lim_a || cnt >= cap - 1 || (has && t_rel < a) || (fire && cur == stop_j);
        // (a lane that is not ready sits the trip out; one whose samples have left the ring needs trip())
        rare = ready ? rare : (pa - 1 < MSKP_BS * (L + MSKP_D - MSKP_NB));
        if (cx.ballot(running && rare) != 0ull || cx.ballot(running && ready) == 0ull)
            return false;
        if (running && ready) {
            if (fire) { // :148-162
                mu = t_val;
                a = t_rel;
                if (mu < 0) {
                    mu++;
                    a--;
                }
                dv = 0;
                om = d_sps;
                cur++;
                tq_front();
            }
            // where the odd iteration reads (:199-201 behind the even one)
            const float m1 = mu + om;
            const float fl1 = floorf(m1);
            const int aO = a + adv_of(fl1);
            const float muO = m1 - fl1;
            // a tag that fires there makes that iteration an even one (:154): it is left for the next trip
            const bool skipO = hasq && (!JOIN || t_rel < ninp_row) && (t_rel >= aO) &&
                               ((float)(t_rel - base_row) < (float)(aO - base_row) + d_sps);
            const unsigned rowE = tap_row(mu), rowO = tap_row(muO);
            const cf yE = fir(a - org, rowE);
            const cf yO = fir(aO - org, rowO);
            worst_row = worst_row > rowE ? worst_row : rowE;
            const cf sqE = cmul_exact(yE, yE);                               // :171
            const cf nlE = cmul_exact(sqE, cconj(cmul_exact(y, y)));         // :173-174
            st8((cf*)(sbuf + (lane >> 4) * 2048 + ((cnt & 15) >> 1) * 256 + (lane & 15) * 16 + (cnt & 1) * 8), yE); // :187
            cnt++;
            if (!skipO) {
                worst_row = worst_row > rowO ? worst_row : rowO;
                const cf sqO = cmul_exact(yO, yO);
                const cf nlO = cmul_exact(sqO, cconj(sqE));
                float err = (nlO - nlE).re;                                  // :178
                err = branchless_clip(err, 3.0f);                            // :179-184
                om += p.gain_omega * err;
                om = d_sps + branchless_clip(om - d_sps, p.limit);
                float m2 = muO + p.gain * err;
                dv += 2;
                y = yO; // :194-196
                nl = nlO;
                m2 += om; // :199-201
                const float fl2 = floorf(m2);
                a = aO + adv_of(fl2);
                mu = m2 - fl2;
            } else {
                dv += 1;
                y = yE;
                nl = nlE;
                a = aO;
                mu = muO;
            }
        }
        return true;
    };

    int flushed = 0; // units: symbols of this lane already written to its staging row
    // NP pairs of iterations with nothing in the way for any running lane (fast_run_pairs() has made
    // sure): no tag can fire, no call or unit can end, the samples are in the ring, mu and omega are
    // in their normal ranges.  No test inside; only the real part of nlin_out is formed on the way
    // (:174, :178), the imaginary part of the last one -- state, d_dly_diff_1 -- at the end.
    int mycan = 0;     // pairs this lane is good for (set by fast_run_pairs; 0: it is not ready, or something is due)
    bool single = false; // some ready lane needs a trip of its own kind (a tag, an end, an odd iteration)
    auto run_pairs = [&](const int NP) {
        if (mycan >= 1) { // (every lane does as many of the NP pairs as it is good for)
            const int mine = mycan < NP ? mycan : NP;
            cf ysq = cmul_exact(y, y);
            cf sqE = ysq, sqO = ysq, yO = y;
            float nlr = nl.re;
            for (int it = 0; it < NP; it++) {
                if (it >= mine)
                    break;
                const float m1 = mu + om;                                    // :199-201 behind the even iteration
                const float fl1 = floorf(m1);
                const int aO = a + (int)fl1;
                const float muO = m1 - fl1;
                const cf yE = fir(a - org, (unsigned)(int)rintf(mu * 128.0f));
                yO = fir(aO - org, (unsigned)(int)rintf(muO * 128.0f));
                sqE = cmul_exact(yE, yE);                                    // :171
                const float nlEr = sqE.re * ysq.re + sqE.im * ysq.im;        // :173-174, real part
                st8((cf*)(sbuf + (lane >> 4) * 2048 + ((cnt & 15) >> 1) * 256 + (lane & 15) * 16 + (cnt & 1) * 8), yE); // :187
                cnt++;
                sqO = cmul_exact(yO, yO);
                nlr = sqO.re * sqE.re + sqO.im * sqE.im;
                const float err = branchless_clip(nlr - nlEr, 3.0f);         // :178-184
                om += p.gain_omega * err;
                om = d_sps + branchless_clip(om - d_sps, p.limit);
                float m2 = muO + p.gain * err;
                m2 += om;
                const float fl2 = floorf(m2);
                a = aO + (int)fl2;
                mu = m2 - fl2;
                ysq = sqO;
            }
            dv += 2 * mine;
            y = yO;
            nl = mk(nlr, sqO.im * sqE.re - sqO.re * sqE.im); // :174 of the last odd iteration
        }
    };
    // the longest run some lane is good for: 8, 4, 2 pairs or none
    auto fast_run_pairs = [&]() -> int {
        const int pa = a - org;
        const bool vis = !JOIN || t_rel < ninp_row;
        const bool has = hasq && vis;
        // items this lane may move before the ring ends / before anything has to be looked at
        const int ring_room = MSKP_BS * L - look - pa;
        int room = 0x3fffffff;
        if (has) {
            const int r = (t_rel - cd - 1) - a;                        // the front tag's window (:142)
            room = room < r ? room : r;
        }
        int prs = 0x3fffffff;                                          // pairs by count
        if (JOIN) {
            const int r = ninp_row - look - a;
            room = room < r ? room : r;
            prs = nout_tot - 1 - cnt;
        } else {
            const int r = lim_a - 1 - a;
            room = room < r ? room : r;
            prs = cap - 2 - cnt;
        }
        // (lanes that have run ahead of the others wait for the ring to move on: they sit the run out,
        // and so does a lane that is good for fewer pairs than the run is long -- the lanes at the back,
        // which hold the ring, have the most room)
        const bool rdy = (pa + look <= MSKP_BS * L) && (pa - 1 >= MSKP_BS * (L + MSKP_D - MSKP_NB));
        const bool plain = ((dv & 1) == 0) && !needq && !(!JOIN && warm) && (mu >= 0.f && mu <= 1.f) && (om >= 0.5f && om <= 30.f);
        int can_evt = 0; // pairs before something is due in this lane
        if (plain && room >= 1) {
            can_evt = (int)((float)(room - 1) * p.padv_inv);
            can_evt = can_evt < prs ? can_evt : prs;
        }
        int can = ring_room >= 1 ? (int)((float)(ring_room - 1) * p.padv_inv) : 0;
        can = can < can_evt ? can : can_evt;
        { // (the symbol stage; a single trip may follow the run)
            const int st = 15 - (cnt - flushed);
            can = can < st ? can : st;
        }
        const bool in = running && rdy;
        mycan = in ? can : 0;
        single = cx.ballot(in && can_evt < 2) != 0ull; // (not the lanes the ring holds back: they wait)
        if (cx.ballot(mycan >= 8) != 0ull)
            return 8;
        if (cx.ballot(mycan >= 4) != 0ull)
            return 4;
        return cx.ballot(mycan >= 2) != 0ull ? 2 : 0;
    };

    // ---- units: symbols leave the stage eight at a time, 64 contiguous bytes per lane, written by the
    // four lanes that serve it (16 cache lines per store instruction instead of 64)
    // symbol j of this lane goes to out_row[j]: the unit's slots of the staging row / the channel's output row
    cf* const out_base = JOIN ? p.syms : p.stage;
    const unsigned stage_off = JOIN ? (unsigned)((long)cc * p.out_stride * 8) : (unsigned)(((long)cc * p.stage_stride + q0) * 8);
    auto drain_syms = [&]() { // (this lane's staged symbols, one by one: before its count jumps, and at the end)
        for (int j = flushed; j < cnt; j++)
            *(cf*)((char*)out_base + stage_off + (unsigned)j * 8u) =
                ld8((const cf*)(sbuf + (lane >> 4) * 2048 + ((j & 15) >> 1) * 256 + (lane & 15) * 16 + (j & 1) * 8));
        flushed = cnt;
    };
    auto flush_syms = [&]() {
        if (JOIN) { // (behind a jump the count may be odd: the 16-byte reads below want pairs)
            const bool odd = valid && (flushed & 1) && cnt > flushed;
            if (cx.ballot(odd) != 0ull) {
                if (odd) {
                    *(cf*)((char*)out_base + stage_off + (unsigned)flushed * 8u) =
                        ld8((const cf*)(sbuf + (lane >> 4) * 2048 + ((flushed & 15) >> 1) * 256 + (lane & 15) * 16 + 8));
                    flushed++;
                }
            }
        }
        const bool full = valid && (cnt - flushed >= 8);
        const u64 fm = cx.ballot(full);
        if (fm == 0ull)
            return;
        const int qq = lane >> 4;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j < ngrp) {
                const int src = 16 * j + (lane & 15);
                const int tf = cx.shfl_i32(flushed, src);
                const unsigned tso = (unsigned)cx.shfl_i32((int)stage_off, src);
                const bool act = ((fm >> src) & 1ull) != 0ull;
                if (act) {
                    cf s0, s1;
                    ld16((const cf*)(sbuf + j * 2048 + ((((tf & 15) >> 1) + qq) & 7) * 256 + (lane & 15) * 16), s0, s1);
                    cx.store16(out_base, tso + (unsigned)(tf + 2 * qq) * 8u, s0, s1);
                }
            }
        }
        if (full)
            flushed += 8;
    };

    // ---- JOIN: a lane stands where restart point `cand`'s tag B is about to fire
    auto walk = [&]() {
        if (!atj)
            return;
        atj = false;
        bool moved = false;
        for (;;) {
            const mskp_res r = rr[cand];
            const bool same = r.kind != MSKP_KIND_NONE && mskp_same_bits(r.ay, y) && mskp_same_bits(r.anl, nl);
            // The unit ran blind to the general_work calls.  Its run is the serial loop's if no tag it
            // used lies outside the call's range (:125-130, the same limit in every call of a step) and
            //  * without max_noutput_items: every iteration top in it passes :138 in the call in progress;
            //  * with max_noutput_items = Q: every call that starts inside it is offered Q outputs
            //    (forecast(Q) fits what is left of the row).  Such a call ends at the top behind its
            //    Q-th symbol and the next one starts there: the loop state is untouched, the tags are
            //    fetched again from nitems_read on -- the front tag is still the front tag (a tag that
            //    reset the loop lies behind iidx one iteration later when d_sps >= 2; a stale tag would
            //    be dropped, but a unit stops at a stale tag).
            bool clean = same && (r.end.a + 1 + cd <= ninp_row) && np < MSKP_SMAX;
            // (a unit that gave up -- at a stale tag, say -- with a call boundary inside: whether that tag was
            // dropped at the boundary or still blocks depends on where exactly the boundary fell: run serially)
            if (p.max_noutput > 0 && r.kind == MSKP_KIND_NEXT)
                clean = clean && msk_forecast(d_sps, p.max_noutput) <= (n - r.end.a) - 1 &&
                        (cnt + r.end.cnt + p.max_noutput <= p.out_cap);
            else
                clean = clean && (cnt + r.end.cnt < nout_tot);
            if (!clean) {
                cand++;
                cand_j = cand < K ? rs[cand].jA + 1 + nc : 0x7fffffff;
                break; // through the junction, serially
            }
            if (!moved)
                drain_syms();
            moved = true;
            mskp_piece pc;
            pc.out0 = cnt;
            pc.src0 = rs[cand].q0;
            pc.cnt = r.end.cnt;
            p.pieces[(long)cc * MSKP_SMAX + np] = pc;
            np++;
            cnt += r.end.cnt;
            while (cnt >= nout_tot) { // (calls that began and ended inside the unit; only with max_noutput_items)
                ototal = nout_tot;
                nout_tot += p.max_noutput;
                base_row = r.end.a; // (somewhere at or before: nothing reads it before the next call starts)
            }
            a = r.end.a;
            mu = r.end.mu;
            om = r.end.omega;
            dv = r.end.div;
            y = r.end.y;
            nl = r.end.nl;
            cur = r.end.cur + nc;
            status |= r.status;
            cand++;
            cand_j = cand < K ? rs[cand].jA + 1 + nc : 0x7fffffff;
            if (r.kind == MSKP_KIND_NEXT && cand < K)
                continue; // that unit ended where the next one's tag B is about to fire
            break;
        }
        running = true;
        jump_to(a); // (nothing was fetched for this lane while it stood at the junction)
        if (moved) {
            flushed = cnt;
            tq_load();
        }
    };

    // ---- the recurrence
    bool all_done = false;
    int since_walk = 0, since_flush = 0;
#define PFB
#define PFE(acc)
    while (!all_done) {
        // block L has arrived when at most the MSKP_D - 1 younger blocks (four transfers each; whatever
        // else is in flight is younger still) are outstanding; then the next one goes out, into the
        // rows of block L + MSKP_D - MSKP_NB, which every lane has left
        { PFB
        cx.template wait_vm<(MSKP_BS / 2) * (MSKP_D - 1)>();
        PFE(pf_wait) }
        L++;
        { PFB
        issue_block(L + MSKP_D - 1);
        PFE(pf_issue) }
        for (;;) {
            // lanes waiting at a junction are served when nobody runs any more, or every
            // MSKP_WALK_EVERY trips (the walk reads records from memory: the whole wave waits)
            const bool idle = cx.ballot(running) == 0ull;
            if (JOIN && (idle || since_walk >= MSKP_WALK_EVERY)) {
                since_walk = 0;
                PFB
                if (cx.ballot(atj) != 0ull)
                    walk();
                PFE(pf_walk)
            }
            if (idle && cx.ballot(running) == 0ull) {
                all_done = true;
                break;
            }
            since_walk++;
            if (cx.ballot(running && ((a - org) - 1 < MSKP_BS * (L + MSKP_D - MSKP_NB + 1))) == 0ull)
                break;
            { PFB
            const int npairs = fast_run_pairs();
            if (npairs == 8) {
                run_pairs(8);
            } else if (npairs == 4)
                run_pairs(4);
            else if (npairs == 2)
                run_pairs(2);
            if (npairs == 0 || single) {
                if (!fast_trip())
                    trip();
            }
            since_flush += npairs + 1;
            PFE(pf_trip) }
            if (since_flush >= 4) { // (at most 7 symbols wait in a lane's stage behind a flush, a run adds 8)
                since_flush = 0;
                flush_syms();
            }
        }
    }
    cx.template wait_vm<0>();

    if (!valid)
        return;
    if (worst_row >= (unsigned)MSK_ZERO_ROW)
        status |= MSK_ST_INTERP_RANGE;
    drain_syms(); // what is left in the stage
    if (!JOIN) {
        mskp_res r;
        r.end.a = a;
        r.end.mu = mu;
        r.end.omega = om;
        r.end.div = dv;
        r.end.y = y;
        r.end.nl = nl;
        r.end.cur = cur;
        r.end.cnt = cnt;
        r.ay = ay;
        r.anl = anl;
        r.kind = kind;
        r.status = status;
        rr[k] = r;
        return;
    }
    // ---- JOIN: the state and the carry of the next call, as k_msk.h leaves them
    p.mu[c] = mu;
    p.omega[c] = om;
    p.div[c] = dv;
    p.dly1[c] = y;
    p.dly2[c] = y;
    p.diff1[c] = nl;
    const int base = base_row + pending; // items consumed
    p.nread[c] = R + (u64)base;
    p.produced[c] = ototal;
    p.consumed[c] = base;
    p.npieces[c] = np;
    cf* cout = p.carry_out + (long)c * p.carry_cap;
    int left = n - base_row; // pending items for the next call
    const int ccap = p.carry_cap < MSK_CARRY_MAX ? p.carry_cap : MSK_CARRY_MAX;
    if (left + 1 > ccap) {
        status |= MSK_ST_CARRY_OVERFLOW;
        left = ccap - 1;
    }
    for (int j = 0; j <= left; j++)
        cout[j] = fetch(base_row - 1 + j);
    p.carry_len_out[c] = left;
    // tags the scheduler still holds: offset >= nitems_read
    {
        int j = cur;
        while (j > 0 && ctl[j - 1].rel >= base_row)
            j--;
        while (j < ntot && ctl[j].rel < base_row)
            j++;
        tag_rec* cto = p.ctag_out + (long)c * p.ctag_cap;
        int w = 0;
        for (; j < ntot; j++) {
            const msk_ctag e = ctl[j];
            if (w < p.ctag_cap) {
                tag_rec tg;
                tg.offset = p.W + (u64)(long long)e.rel;
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
    }
    p.status[c] = status;
}

// symbols of the accepted units: staging row -> output row.  Workgroup (x, channel) copies its
// share of every piece.
template <class Ctx>
AISX_DI void mskp_gather_body(Ctx& cx, const MskpGatherParams& p)
{
    const int c = cx.by();
    const int np = p.npieces[c];
    const mskp_piece* pc = p.pieces + (long)c * MSKP_SMAX;
    const cf* src = p.stage + (long)c * p.stage_stride;
    cf* dst = p.syms + (long)c * p.out_stride;
    for (int i = cx.bx(); i < np; i += MSKP_GATHER_X) {
        const mskp_piece e = pc[i];
        for (int j = cx.tid(); j < e.cnt; j += cx.nthreads())
            dst[e.out0 + j] = src[e.src0 + j];
    }
}

} // namespace aisx
