// k_msk.h -- msk_timing_recovery_cc (reference: lib/msk_timing_recovery_cc_impl.cc
// :107-206) with the NRZI bit tail of python/ais_demod.py:48-52 + lib/invert_impl.cc
// :62-64 fused into the epilogue.  One lane per channel: the loop is a strict
// recurrence through (mu, omega, iidx), so the only parallelism is across
// channels.  All arithmetic is the reference's float/double sequence, unfused,
// so the symbols are bit-identical to the oracle's.
#pragma once
#include "aisx_common.h"

namespace aisx {

enum { MSK_ST_INTERP_RANGE = 1, MSK_ST_CARRY_OVERFLOW = 2, MSK_ST_TAGCARRY_OVERFLOW = 4, MSK_ST_OUT_FULL = 8 };

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

template <class Ctx>
AISX_DI void msk_body(Ctx& cx, const MskParams& p)
{
    const int c = cx.bx() * cx.nthreads() + cx.tid();
    if (c >= p.nchan)
        return;
    const float d_sps = p.d_sps;
    float d_mu = p.mu[c], d_omega = p.omega[c];
    int d_div = p.div[c];
    cf d_dly_conj_1 = p.dly1[c], d_dly_conj_2 = p.dly2[c], d_dly_diff_1 = p.diff1[c];
    cf tprev = p.tail_prev_sym[c];
    unsigned char tbit = p.tail_prev_bit[c];
    const unsigned long long R = p.nread[c];
    int status = 0;

    const cf* cin = p.carry_in + (long)c * p.carry_cap; // [0] = item before R, then pending
    const int pending = p.carry_len_in[c];
    const cf* xin = p.in + (long)c * p.in_stride;
    const int navail = pending + p.n;
    // item idx of the items on offer, idx in [-1, navail)
    auto fetch = [&](int idx) -> cf {
        const int q = idx + 1;
        return (q <= pending) ? cin[q] : xin[q - 1 - pending];
    };

    // logical tag list = carried tags, then this call's tags
    const tag_rec* ctg = p.ctag_in + (long)c * p.ctag_cap;
    const int nct = p.ctag_n_in[c];
    const tag_rec* ntg = p.tags ? p.tags + (long)c * p.tag_cap : nullptr;
    int nnt = p.tags ? p.tag_count[c] : 0;
    if (nnt > p.tag_cap)
        nnt = p.tag_cap;
    const int ntot = nct + nnt;
    int tpos = 0;
    auto tag_at = [&](int k) -> const tag_rec& { return (k < nct) ? ctg[k] : ntg[k - nct]; };
    auto skip_other_keys = [&]() {
        while (tpos < ntot && tag_at(tpos).key != KEY_TIME_EST)
            tpos++;
    };

    cf* osym = p.syms ? p.syms + (long)c * p.out_stride : nullptr;
    float* oerr = p.err ? p.err + (long)c * p.out_stride : nullptr;
    float* omu = p.mu_out ? p.mu_out + (long)c * p.out_stride : nullptr;
    unsigned char* obit = p.bits ? p.bits + (long)c * p.out_stride : nullptr;

    // Stream mode plays the scheduler: general_work() is called again and again on
    // what is left until forecast(1) no longer fits.  `base` = items consumed and
    // `ototal` = items produced by the calls made so far in this launch.
    int base = 0, ototal = 0;
    for (;;) {
        const unsigned long long Rc = R + (unsigned long long)base; // nitems_read(0) of this call
        int ninput, noutput;
        if (p.stream_mode) {
            ninput = (navail - base) - 1; // one look-ahead item is kept out of sight
            noutput = 0;
            if (ninput > 0) {
                noutput = (int)((ninput - 3.0 * d_sps - 8) / (2.0 * d_sps)) + 2;
                while (noutput > 0 && msk_forecast(d_sps, noutput) > ninput)
                    noutput--;
            }
            if (noutput > p.out_cap - ototal) {
                noutput = p.out_cap - ototal;
                status |= MSK_ST_OUT_FULL;
            }
        } else {
            ninput = p.gr_ninput;
            noutput = p.gr_noutput;
        }
        const int ninp = (int)(ninput - 3.0 * d_sps); // :119
        if (ninp <= 0 || noutput <= 0)
            break;
        // get_tags_in_range(nitems_read, nitems_read + ninp, "time_est") (:125-130)
        tpos = 0;
        skip_other_keys();
        while (tpos < ntot && tag_at(tpos).offset < Rc) {
            tpos++;
            skip_other_keys();
        }
        const unsigned long long rend = Rc + (unsigned long long)ninp;
        int oidx = 0, iidx = 0;
        float err_out = 0;
        while (oidx < noutput && iidx < ninp) { // :138
            if (tpos < ntot && tag_at(tpos).offset < rend) { // tags.size() > 0
                const int offset = (int)(tag_at(tpos).offset - Rc);
                if ((offset >= iidx) && ((float)offset < ((float)iidx + d_sps))) { // :142
                    const float center = (float)tag_at(tpos).value;
                    if (center != center) { // NaN :144-147
                        tpos++;
                        skip_other_keys();
                    } else {
                        d_mu = center;
                        iidx = offset;
                        if (d_mu < 0) {
                            d_mu++;
                            iidx--;
                        }
                        d_div = 0;
                        d_omega = d_sps;
                        d_dly_conj_2 = d_dly_conj_1;
                        tpos++;
                        skip_other_keys();
                    }
                }
            }
            // mmse_fir_interpolator_cc::interpolate(&in[iidx], d_mu) (:170)
            const int imu = (int)rint(d_mu * 128.0f);
            cf in_interp = mk(0.f, 0.f);
            if (imu < 0 || imu > 128) {
                status |= MSK_ST_INTERP_RANGE; // upstream throws std::runtime_error
            } else {
                const float* tp = p.mmse + imu * 8;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const cf s = fetch(base + iidx + k);
                    const float tk = tp[7 - k];
                    in_interp.re += s.re * tk;
                    in_interp.im += s.im * tk;
                }
            }
            const cf sq = cmul_exact(in_interp, in_interp);                      // :171
            const cf dly_conj = cconj(cmul_exact(d_dly_conj_2, d_dly_conj_2));   // :173
            const cf nlin_out = cmul_exact(sq, dly_conj);                        // :174
            err_out = (nlin_out - d_dly_diff_1).re;                              // :178
            if (d_div % 2) {                                                     // :179-184
                err_out = branchless_clip(err_out, 3.0f);
                d_omega += p.gain_omega * err_out;
                d_omega = d_sps + branchless_clip(d_omega - d_sps, p.limit);
                d_mu += p.gain * err_out;
            }
            if (!(d_div % 2) || p.osps == 2) { // :186-191
                const int oo = ototal + oidx;
                if (osym)
                    osym[oo] = in_interp;
                if (oerr)
                    oerr[oo] = err_out;
                if (omu)
                    omu[oo] = d_mu;
                if (obit) {
                    // quadrature_demod_cf(pi/2) -> binary_slicer_fb -> diff_decoder_bb(2) -> invert
                    const cf prod = cmul_exact(in_interp, cconj(tprev));
                    const float fm = 1.57079632679489661923f * fast_atan2f_tab(prod.im, prod.re, p.atan_tab);
                    const unsigned char b = fm >= 0 ? 1 : 0;
                    const unsigned char d = (unsigned char)(((unsigned)(b - tbit)) % 2u);
                    obit[oo] = (unsigned char)((d ^ 0x01) & 0x01);
                    tprev = in_interp;
                    tbit = b;
                }
                oidx++;
            }
            d_div++;
            d_dly_conj_1 = in_interp; // :194-196
            d_dly_conj_2 = d_dly_conj_1;
            d_dly_diff_1 = nlin_out;
            d_mu += d_omega; // :199-201
            const float fl = floorf(d_mu);
            iidx += (int)fl;
            d_mu = d_mu - fl;
        }
        base += iidx; // consume_each(iidx)
        ototal += oidx;
        if (!p.stream_mode || (iidx <= 0 && oidx == 0))
            break;
    }
    const int iidx = base, oidx = ototal;
    // consume_each(iidx)
    p.mu[c] = d_mu;
    p.omega[c] = d_omega;
    p.div[c] = d_div;
    p.dly1[c] = d_dly_conj_1;
    p.dly2[c] = d_dly_conj_2;
    p.diff1[c] = d_dly_diff_1;
    p.tail_prev_sym[c] = tprev;
    p.tail_prev_bit[c] = tbit;
    const unsigned long long Rn = R + (unsigned long long)iidx;
    p.nread[c] = Rn;
    p.produced[c] = oidx;
    p.consumed[c] = iidx;

    cf* cout = p.carry_out + (long)c * p.carry_cap;
    if (p.stream_mode) {
        int left = navail - iidx; // pending items for the next call
        if (left + 1 > p.carry_cap) {
            status |= MSK_ST_CARRY_OVERFLOW;
            left = p.carry_cap - 1;
        }
        for (int k = 0; k <= left; k++)
            cout[k] = fetch(iidx - 1 + k);
        p.carry_len_out[c] = left;
        // tags the scheduler still holds: offset >= nitems_read
        tag_rec* cto = p.ctag_out + (long)c * p.ctag_cap;
        int w = 0;
        for (int k = 0; k < ntot; k++) {
            const tag_rec& tg = tag_at(k);
            if (tg.key != KEY_TIME_EST || tg.offset < Rn)
                continue;
            if (w < p.ctag_cap)
                cto[w] = tg;
            else
                status |= MSK_ST_TAGCARRY_OVERFLOW;
            w++;
        }
        p.ctag_n_out[c] = w < p.ctag_cap ? w : p.ctag_cap;
    } else {
        cout[0] = (iidx > 0) ? fetch(iidx - 1) : cin[0];
        p.carry_len_out[c] = 0;
        p.ctag_n_out[c] = 0;
    }
    p.status[c] = status;
}

} // namespace aisx
