// k_msk.h -- msk_timing_recovery_cc (reference: lib/msk_timing_recovery_cc_impl.cc
// :107-206) with the NRZI bit tail of python/ais_demod.py:48-52 + lib/invert_impl.cc
// :62-64 fused into the epilogue.
//
// The loop is a strict recurrence through (mu, omega, iidx): the only parallelism
// is across channels.  One lane owns MSK_NCH = 2 channels and walks both
// recurrences in the same instruction stream: a lone wave per SIMD can only
// issue a dependent instruction every few cycles, two independent chains fill
// those bubbles (and each chain's LDS latency hides behind the other).  Every
// channel keeps a ring of its last 128 samples in LDS, slot-major
// (ring[slot][lane]): a lane always touches its own pair of banks, so the 8-tap
// reads are conflict free however far the lanes drift apart.  The memory schedule
// is the same for all lanes: chunk t = new samples [64t, 64t+64) of every channel
// is fetched (each lane walking its own rows) BEFORE the iterations that consume
// chunk t-1, so the loads fly under the recurrence, and landed afterwards; lanes
// then iterate, each channel at its own pace, until none can go on without the
// next chunk.  Symbols and bits are stored straight from the loop.  All
// arithmetic is the reference's float/double sequence, unfused: bit-identical to
// the CPU restatement.
#pragma once
#include "aisx_common.h"

namespace aisx {

enum { MSK_ST_INTERP_RANGE = 1, MSK_ST_CARRY_OVERFLOW = 2, MSK_ST_TAGCARRY_OVERFLOW = 4, MSK_ST_OUT_FULL = 8 };

constexpr int MSK_T = 64;
constexpr int MSK_NCH = 2;      // channels per lane
constexpr int MSK_RING = 128;   // slots per channel (power of two)
constexpr int MSK_SLOTS = MSK_RING + 8; // + 8 mirror slots: an 8-tap read never wraps
constexpr int MSK_CHUNK = 64;   // samples per chunk
constexpr int MSK_OFF = 64;     // ring slot of new-sample index s is (s + MSK_OFF) & 127
constexpr int MSK_CARRY_MAX = 60;
constexpr int MSK_TAPS_PITCH = 9; // floats per table row in LDS (8 taps + 1: spreads rows over banks)
constexpr int MSK_LDS_RING1 = MSK_SLOTS * 64 * 8; // one channel set
constexpr int MSK_LDS_RING = MSK_NCH * MSK_LDS_RING1;
constexpr int MSK_LDS_MMSE = ((129 * MSK_TAPS_PITCH * 4 + 15) / 16) * 16;
constexpr int MSK_LDS_ATAN = 260 * 4;
constexpr int MSK_LDS_BYTES = MSK_LDS_RING + MSK_LDS_MMSE + MSK_LDS_ATAN;

struct MskParams {
    int nchan;
    // loop constants (set_sps / set_gain / set_limit, :69-96)
    float d_sps, gain, gain_omega, limit;
    int osps;
    // per-channel loop state
    float* mu; float* omega; int* div;
    cf* dly1; cf* dly2; cf* diff1;
    cf* tail_prev_sym; unsigned char* tail_prev_bit;
    unsigned long long* nread; // nitems_read(0)
    // input: stream mode = carry (pre-item + pending items) followed by n new items
    const cf* in; long in_stride; int n;
    const cf* carry_in; cf* carry_out; const int* carry_len_in; int* carry_len_out; int carry_cap;
    // GNU Radio mode (stream_mode == 0): explicit ninput/noutput, nothing carried but the pre-item
    int stream_mode; int gr_ninput; int gr_noutput;
    // tags: new ones from this call + carried ones
    const tag_rec* tags; const int* tag_count; int tag_cap;
    const tag_rec* ctag_in; tag_rec* ctag_out; const int* ctag_n_in; int* ctag_n_out; int ctag_cap;
    // outputs
    cf* syms; float* err; float* mu_out; unsigned char* bits; long out_stride; int out_cap;
    int* produced; int* consumed; int* status;
    const float* mmse; // [129][8]
    const float* atan_tab;
};

AISX_HD int msk_forecast(float d_sps, int noutput_items)
{
    return (int)ceil((noutput_items * d_sps * 2) + 3.0 * d_sps + 8u);
}


// everything one channel's recurrence carries
struct MskChan {
    bool live, done;
    int cc;
    float d_mu, d_omega;
    int d_div;
    cf dly1, dly2, diff1, prev_sq, tprev;
    unsigned char tbit;
    unsigned long long R, Rc, rend;
    int status, pending, navail;
    const tag_rec *ctg, *ntg;
    int nct, ntot, tpos;
    unsigned long long nt_off;
    float nt_val;
    int nt_rel;
    float *oerr, *omu;
    cf* osymg;
    unsigned char* obitg;
    const cf* myin;
    cf* myring;
    int base, ototal, iidx, oidx, ninp, noutput;
};

AISX_DI const tag_rec& msk_tag_at(const MskChan& c, int k) { return (k < c.nct) ? c.ctg[k] : c.ntg[k - c.nct]; }

// advance to the next time_est tag and keep its offset/value in registers: the loop
// tests the front tag on every iteration and must not pay a global load for that
AISX_DI void msk_skip_other_keys(MskChan& c)
{
    while (c.tpos < c.ntot && msk_tag_at(c, c.tpos).key != KEY_TIME_EST)
        c.tpos++;
    if (c.tpos < c.ntot) {
        c.nt_off = msk_tag_at(c, c.tpos).offset;
        c.nt_val = (float)msk_tag_at(c, c.tpos).value;
    } else {
        c.nt_off = ~0ull;
    }
}

// "scheduler": set up the next general_work() call of this channel (stream mode) or
// the one call of the GNU Radio mode
AISX_DI void msk_setup_round(MskChan& c, const MskParams& p)
{
    const float d_sps = p.d_sps;
    int ninput;
    if (p.stream_mode) {
        ninput = (c.navail - c.base) - 1; // one look-ahead item is kept out of sight
        c.noutput = 0;
        if (ninput > 0) {
            c.noutput = (int)((ninput - 3.0 * d_sps - 8) / (2.0 * d_sps)) + 2;
            while (c.noutput > 0 && msk_forecast(d_sps, c.noutput) > ninput)
                c.noutput--;
        }
        if (c.noutput > p.out_cap - c.ototal) {
            c.noutput = p.out_cap - c.ototal;
            c.status |= MSK_ST_OUT_FULL;
        }
    } else {
        ninput = p.gr_ninput;
        c.noutput = p.gr_noutput;
    }
    c.ninp = (int)(ninput - 3.0 * d_sps); // :119
    c.iidx = 0;
    c.oidx = 0;
    if (c.ninp <= 0 || c.noutput <= 0) {
        c.done = true;
        return;
    }
    // get_tags_in_range(nitems_read, nitems_read + ninp, "time_est") (:125-130)
    c.Rc = c.R + (unsigned long long)c.base;
    c.rend = c.Rc + (unsigned long long)c.ninp;
    c.tpos = 0;
    msk_skip_other_keys(c);
    while (c.nt_off < c.Rc) {
        c.tpos++;
        msk_skip_other_keys(c);
    }
    c.nt_rel = (c.nt_off < c.rend) ? (int)(c.nt_off - c.Rc) : 0x7fffffff;
}

// one loop iteration (:170-201)
AISX_DI void msk_iterate(MskChan& c, const MskParams& p, const float* mm, const float* at)
{
    const float d_sps = p.d_sps;
    // mmse_fir_interpolator_cc::interpolate(&in[iidx], d_mu) (:170)
    const int imu = (int)rintf(c.d_mu * 128.0f);
    cf in_interp = mk(0.f, 0.f);
    if (imu < 0 || imu > 128) {
        c.status |= MSK_ST_INTERP_RANGE; // upstream throws std::runtime_error
    } else {
        const float* tp = mm + imu * MSK_TAPS_PITCH;
        const cf* sp = c.myring + ((c.base + c.iidx - c.pending + MSK_OFF) & (MSK_RING - 1)) * 64;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const cf s = sp[k * 64]; // mirror slots: no wrap inside the 8 taps
            const float tk = tp[7 - k];
            in_interp.re += s.re * tk;
            in_interp.im += s.im * tk;
        }
    }
    const cf sq = cmul_exact(in_interp, in_interp); // :171
    // :173 conj(d_dly_conj_2^2): d_dly_conj_2 is always the previous in_interp
    // (:194-195, also after a tag reset :160), so its square is the previous sq
    const cf dly_conj = cconj(c.prev_sq);
    const cf nlin_out = cmul_exact(sq, dly_conj); // :174
    float err_out = (nlin_out - c.diff1).re;        // :178
    if (c.d_div & 1) {                              // :179-184
        err_out = branchless_clip(err_out, 3.0f);
        c.d_omega += p.gain_omega * err_out;
        c.d_omega = d_sps + branchless_clip(c.d_omega - d_sps, p.limit);
        c.d_mu += p.gain * err_out;
    }
    if (!(c.d_div & 1) || p.osps == 2) { // :186-191
        const int oo = c.ototal + c.oidx;
        if (c.osymg)
            c.osymg[oo] = in_interp;
        if (c.oerr)
            c.oerr[oo] = err_out;
        if (c.omu)
            c.omu[oo] = c.d_mu;
        // quadrature_demod_cf(pi/2) -> binary_slicer_fb -> diff_decoder_bb(2) -> invert
        const cf prod = cmul_exact(in_interp, cconj(c.tprev));
        const float fm = 1.57079632679489661923f * fast_atan2f_tab(prod.im, prod.re, at);
        const unsigned char b = fm >= 0 ? 1 : 0;
        const unsigned char d = (unsigned char)(((unsigned)(b - c.tbit)) % 2u);
        if (c.obitg)
            c.obitg[oo] = (unsigned char)((d ^ 0x01) & 0x01);
        c.tprev = in_interp;
        c.tbit = b;
        c.oidx++;
    }
    c.d_div++;
    c.dly1 = in_interp; // :194-196
    c.dly2 = c.dly1;
    c.prev_sq = sq;
    c.diff1 = nlin_out;
    c.d_mu += c.d_omega; // :199-201
    const float fl = floorf(c.d_mu);
    c.iidx += (int)fl;
    c.d_mu = c.d_mu - fl;
}

template <class Ctx>
AISX_DI void msk_body(Ctx& cx, const MskParams& p)
{
    const int l = cx.tid();
    char* lds = cx.lds();
    float* mm = (float*)(lds + MSK_LDS_RING);
    float* at = (float*)(lds + MSK_LDS_RING + MSK_LDS_MMSE);
    for (int i = l; i < 129 * 8; i += 64)
        mm[(i >> 3) * MSK_TAPS_PITCH + (i & 7)] = p.mmse[i];
    for (int i = l; i < 257; i += 64)
        at[i] = p.atan_tab[i];

    const float d_sps = p.d_sps;
    const int n = p.n;
    MskChan ch[MSK_NCH];
#pragma unroll
    for (int u = 0; u < MSK_NCH; u++) {
        MskChan& c = ch[u];
        const int cidx = (cx.bx() * MSK_NCH + u) * 64 + l;
        c.live = cidx < p.nchan;
        c.cc = c.live ? cidx : (p.nchan - 1); // dead lanes mirror the last channel read-only
        c.myring = (cf*)(lds + u * MSK_LDS_RING1) + l; // slot k of this channel: myring[k * 64]
        c.d_mu = p.mu[c.cc];
        c.d_omega = p.omega[c.cc];
        c.d_div = p.div[c.cc];
        c.dly1 = p.dly1[c.cc];
        c.dly2 = p.dly2[c.cc];
        c.diff1 = p.diff1[c.cc];
        c.prev_sq = cmul_exact(c.dly2, c.dly2);
        c.tprev = p.tail_prev_sym[c.cc];
        c.tbit = p.tail_prev_bit[c.cc];
        c.R = p.nread[c.cc];
        c.Rc = c.R;
        c.rend = c.R;
        c.status = 0;
        // items on offer: logical index q in [-1, navail): q = -1 the item before
        // nitems_read, then the `pending` carried items, then the n new ones.  The new
        // sample index is s = q - pending; ring slot of s is (s + MSK_OFF) & 127.
        const cf* cin = p.carry_in + (long)c.cc * p.carry_cap;
        c.pending = p.carry_len_in[c.cc];
        if (c.pending > MSK_CARRY_MAX)
            c.pending = MSK_CARRY_MAX;
        c.navail = c.pending + n;
        for (int q = -1; q < c.pending; q++) {
            const int slot = (q - c.pending + MSK_OFF) & (MSK_RING - 1);
            c.myring[slot * 64] = cin[q + 1];
            if (slot < 8)
                c.myring[(MSK_RING + slot) * 64] = cin[q + 1];
        }
        // logical tag list = carried tags, then this call's tags
        c.ctg = p.ctag_in + (long)c.cc * p.ctag_cap;
        c.nct = p.ctag_n_in[c.cc];
        c.ntg = p.tags ? p.tags + (long)c.cc * p.tag_cap : nullptr;
        int nnt = p.tags ? p.tag_count[c.cc] : 0;
        if (nnt > p.tag_cap)
            nnt = p.tag_cap;
        c.ntot = c.nct + nnt;
        c.tpos = 0;
        c.nt_off = ~0ull;
        c.nt_val = 0.f;
        c.nt_rel = 0x7fffffff;
        c.oerr = p.err ? p.err + (long)c.cc * p.out_stride : nullptr;
        c.omu = p.mu_out ? p.mu_out + (long)c.cc * p.out_stride : nullptr;
        c.osymg = p.syms ? p.syms + (long)c.cc * p.out_stride : nullptr;
        c.obitg = p.bits ? p.bits + (long)c.cc * p.out_stride : nullptr;
        c.myin = p.in + (long)c.cc * p.in_stride;
        c.base = 0;
        c.ototal = 0;
        c.iidx = 0;
        c.oidx = 0;
        c.ninp = 0;
        c.noutput = 0;
        c.done = !c.live;
        if (!c.done)
            msk_setup_round(c, p);
    }
    const int jump_margin = (int)ceilf(d_sps) + 1; // a tag may move iidx forward by < d_sps

    // chunks cover the new samples plus an 8-sample zero guard: the reference's loop
    // bound lets the interpolator look a few items past ninput_items when sps < 4;
    // here those items read as zero
    const int nchunks = (n + 8 + MSK_CHUNK - 1) / MSK_CHUNK;
    // chunk fetch: lane l walks its own channel rows (64 samples = 512 contiguous bytes
    // per channel and chunk; the lines involved stay in L1 across the load instructions)
    cf r[MSK_NCH][MSK_CHUNK];
    auto issue_chunk = [&](int t) {
        const int s0 = t * MSK_CHUNK;
#pragma unroll
        for (int u = 0; u < MSK_NCH; u++)
#pragma unroll
            for (int k = 0; k < MSK_CHUNK; k++) {
                r[u][k] = mk(0.f, 0.f);
                if (ch[u].live && s0 + k < n)
                    r[u][k] = ch[u].myin[s0 + k];
            }
    };
    auto land_chunk = [&](int t) {
        const int slot0 = (t * MSK_CHUNK + MSK_OFF) & (MSK_RING - 1); // multiple of 64
#pragma unroll
        for (int u = 0; u < MSK_NCH; u++) {
#pragma unroll
            for (int k = 0; k < MSK_CHUNK; k++)
                ch[u].myring[(slot0 + k) * 64] = r[u][k];
            if (slot0 == 0) { // mirror the first 8 slots behind the last one
#pragma unroll
                for (int k = 0; k < 8; k++)
                    ch[u].myring[(MSK_RING + k) * 64] = r[u][k];
            }
        }
    };
    issue_chunk(0);
    land_chunk(0);
    int landed = 1; // chunks in the rings
    cx.sync();

    for (;;) {
        if (cx.ballot(!ch[0].done || !ch[1].done) == 0ull)
            break;
        const bool more = landed < nchunks;
        if (more)
            issue_chunk(landed);
        const int loaded_s = landed * MSK_CHUNK; // new samples [.., loaded_s) are in the rings
        // ---------------- the recurrences: every channel goes as far as its data allows ----------------
        for (;;) {
            // (rare) a general_work() call is over (:138): consume, start the next one
            bool rev[MSK_NCH], can[MSK_NCH], tev[MSK_NCH];
#pragma unroll
            for (int u = 0; u < MSK_NCH; u++)
                rev[u] = !ch[u].done && !(ch[u].oidx < ch[u].noutput && ch[u].iidx < ch[u].ninp);
            if (cx.ballot(rev[0] || rev[1]) != 0ull) {
#pragma unroll
                for (int u = 0; u < MSK_NCH; u++)
                    if (rev[u]) {
                        MskChan& c = ch[u];
                        c.base += c.iidx; // consume_each(iidx)
                        c.ototal += c.oidx;
                        const bool progress = (c.iidx > 0) || (c.oidx > 0);
                        if (!p.stream_mode || !progress)
                            c.done = true;
                        else
                            msk_setup_round(c, p);
                    }
            }
#pragma unroll
            for (int u = 0; u < MSK_NCH; u++) {
                const int pos_s = ch[u].base + ch[u].iidx - ch[u].pending;
                can[u] = !ch[u].done && (!more || (pos_s + 8 + jump_margin <= loaded_s));
            }
            if (cx.ballot(can[0] || can[1]) == 0ull)
                break;
            // (rare) a time_est tag lands in [iidx, iidx + d_sps) (:140-164)
#pragma unroll
            for (int u = 0; u < MSK_NCH; u++)
                tev[u] = can[u] && (ch[u].nt_rel >= ch[u].iidx) && ((float)ch[u].nt_rel < ((float)ch[u].iidx + d_sps));
            if (cx.ballot(tev[0] || tev[1]) != 0ull) {
#pragma unroll
                for (int u = 0; u < MSK_NCH; u++)
                    if (tev[u]) {
                        MskChan& c = ch[u];
                        const float center = c.nt_val;
                        if (center == center) { // not NaN (:144-147)
                            c.d_mu = center;
                            c.iidx = c.nt_rel;
                            if (c.d_mu < 0) {
                                c.d_mu++;
                                c.iidx--;
                            }
                            c.d_div = 0;
                            c.d_omega = d_sps;
                            c.dly2 = c.dly1; // (prev_sq already is dly1^2)
                        }
                        c.tpos++;
                        msk_skip_other_keys(c);
                        c.nt_rel = (c.nt_off < c.rend) ? (int)(c.nt_off - c.Rc) : 0x7fffffff;
                    }
            }
#pragma unroll
            for (int u = 0; u < MSK_NCH; u++)
                if (can[u])
                    msk_iterate(ch[u], p, mm, at);
        }
        // ---------------- land the prefetched chunk ----------------
        if (more) {
            land_chunk(landed);
            landed++;
        }
        cx.sync();
    }

#pragma unroll
    for (int u = 0; u < MSK_NCH; u++) {
        MskChan& c = ch[u];
        if (!c.live)
            continue;
        const int cidx = c.cc;
        p.mu[cidx] = c.d_mu;
        p.omega[cidx] = c.d_omega;
        p.div[cidx] = c.d_div;
        p.dly1[cidx] = c.dly1;
        p.dly2[cidx] = c.dly2;
        p.diff1[cidx] = c.diff1;
        p.tail_prev_sym[cidx] = c.tprev;
        p.tail_prev_bit[cidx] = c.tbit;
        const unsigned long long Rn = c.R + (unsigned long long)c.base;
        p.nread[cidx] = Rn;
        p.produced[cidx] = c.ototal;
        p.consumed[cidx] = c.base;
        cf* cout = p.carry_out + (long)cidx * p.carry_cap;
        if (p.stream_mode) {
            int left = c.navail - c.base; // pending items for the next call
            const int cap = p.carry_cap < MSK_CARRY_MAX ? p.carry_cap : MSK_CARRY_MAX;
            if (left + 1 > cap) {
                c.status |= MSK_ST_CARRY_OVERFLOW;
                left = cap - 1;
            }
            for (int k = 0; k <= left; k++)
                cout[k] = c.myring[((c.base - 1 + k - c.pending + MSK_OFF) & (MSK_RING - 1)) * 64];
            p.carry_len_out[cidx] = left;
            // tags the scheduler still holds: offset >= nitems_read
            tag_rec* cto = p.ctag_out + (long)cidx * p.ctag_cap;
            int w = 0;
            for (int k = 0; k < c.ntot; k++) {
                const tag_rec& tg = msk_tag_at(c, k);
                if (tg.key != KEY_TIME_EST || tg.offset < Rn)
                    continue;
                if (w < p.ctag_cap)
                    cto[w] = tg;
                else
                    c.status |= MSK_ST_TAGCARRY_OVERFLOW;
                w++;
            }
            p.ctag_n_out[cidx] = w < p.ctag_cap ? w : p.ctag_cap;
        } else {
            cout[0] = c.myring[((c.base - 1 - c.pending + MSK_OFF) & (MSK_RING - 1)) * 64];
            p.carry_len_out[cidx] = 0;
            p.ctag_n_out[cidx] = 0;
        }
        p.status[cidx] = c.status;
    }
}

} // namespace aisx
