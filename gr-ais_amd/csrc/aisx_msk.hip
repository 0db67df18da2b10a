// aisx_msk.hip -- C ABI (include/aisx.h) for msk_timing_recovery_cc and the NRZI bit tail,
// plus the __global__ wrappers that run the kernel bodies of k_msk.h on gfx950.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "aisx_devctx.h"
#include "aisx_host.h"
#include "aisx_plan.h"
#include "aisx_tables.h"
#include "k_msk.h"
#include "k_mskp.h"

using namespace aisx;

template <bool AUX, bool OSPS2, int LPW>
__global__ __launch_bounds__(64 * msk_waves(LPW)) void k_msk(MskParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
    // the recurrence is latency-bound and issues little: its waves go first on their SIMDs, ahead
    // of the throughput kernels of the other stream that share them
#ifndef MSK_PRIO
#define MSK_PRIO 3
#endif
    __builtin_amdgcn_s_setprio(MSK_PRIO);
    msk_body<DevCtx, AUX, OSPS2, LPW>(cx, p);
}

// the same kernel as the join of the time-parallel recovery (MskParams::ff)
template <int LPW>
__global__ __launch_bounds__(64 * msk_waves(LPW)) void k_msk_ff(MskParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
    __builtin_amdgcn_s_setprio(MSK_PRIO);
    msk_body<DevCtx, false, false, LPW, true>(cx, p);
}

__global__ __launch_bounds__(BT_T) void k_bittail(BitTailParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[260 * 4];
    DevCtx cx{ smem };
    bittail_body(cx, p);
}

// One wave that does nothing for `ticks` periods of the 100 MHz wall clock: queued on a stream
// behind the event of aisx_msk_wait_prepass it gives the recovery kernel, which becomes ready
// at that same event on its own stream, a head start at the dispatcher (see aisx_msk_wait_prepass).
__global__ __launch_bounds__(64) void k_msk_headstart(unsigned ticks)
{
    const unsigned long long t0 = wall_clock64();
    // (bounded whatever the clock does: 2048 sleeps of 32 x 64 cycles are ~2 ms)
    for (int k = 0; k < 2048 && wall_clock64() - t0 < ticks; k++)
        __builtin_amdgcn_s_sleep(32);
}

__global__ __launch_bounds__(256) void k_msk_tagprep(TagPrepParams p)
{
    DevCtx cx{ nullptr };
    tagprep_body(cx, p);
}

// ---- the time-parallel recovery (k_mskp.h): prepass, units, join, gather
__global__ __launch_bounds__(64) void k_mskp_prep(MskpPrepParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[MSKP_PREP_LDS_TAGS * 8];
    DevCtx cx{ smem };
    mskp_prep_body(cx, p);
}
__global__ __launch_bounds__(64) void k_mskp_units(MskpParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
    mskp_body<DevCtx, false>(cx, p);
}
__global__ __launch_bounds__(64) void k_mskp_join(MskpParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
    // one lane per channel, a recurrence: its waves go first on their SIMDs
    __builtin_amdgcn_s_setprio(3);
    mskp_body<DevCtx, true>(cx, p);
}
__global__ __launch_bounds__(256) void k_mskp_gather(MskpGatherParams p)
{
    DevCtx cx{ nullptr };
    mskp_gather_body(cx, p);
}

// launch the timing-recovery build for (err/mu ports connected, osps == 2, channels per wave)
static int msk_launch(const MskParams& p, int nwg, hipStream_t st)
{
    typedef void (*kfn)(MskParams);
    static const kfn fns[20] = {
        k_msk<false, false, 16>, k_msk<false, true, 16>, k_msk<true, false, 16>, k_msk<true, true, 16>,
        k_msk<false, false, 32>, k_msk<false, true, 32>, k_msk<true, false, 32>, k_msk<true, true, 32>,
        k_msk<false, false, 64>, k_msk<false, true, 64>, k_msk<true, false, 64>, k_msk<true, true, 64>,
        k_msk<false, false, 8>,  k_msk<false, true, 8>,  k_msk<true, false, 8>,  k_msk<true, true, 8>,
        k_msk<false, false, 4>,  k_msk<false, true, 4>,  k_msk<true, false, 4>,  k_msk<true, true, 4>,
    };
    static bool big_lds[20] = { false };
    const int li = p.lpw == 16 ? 0 : (p.lpw == 32 ? 1 : (p.lpw == 64 ? 2 : (p.lpw == 8 ? 3 : 4)));
    const int v = li * 4 + (((p.err || p.mu_out) ? 2 : 0) | (p.osps == 2 ? 1 : 0));
    // LDS beyond what the kernel uses keeps other streams' workgroups off this CU: a knob for how
    // much of its SIMDs' issue the recurrence shares (AISX_MSK_LDS_PAD, KiB; experiments)
    static const int pad = [] {
        const char* e = exp_env("AISX_MSK_LDS_PAD");
        return e ? atoi(e) * 1024 : 0;
    }();
    const int lds = std::min(msk_lds_bytes(p.lpw) + pad, 160 * 1024);
    kfn fn = fns[v];
    bool* big = &big_lds[v];
    if (p.ff) { // the join of the time-parallel recovery: a build of its own (stream mode, osps 1, no err / mu)
        static const kfn ffs[5] = { k_msk_ff<16>, k_msk_ff<32>, k_msk_ff<64>, k_msk_ff<8>, k_msk_ff<4> };
        static bool big_ff[5] = { false };
        if (p.err || p.mu_out || p.osps == 2) {
            set_err("msk_launch: the join build has no err / mu ports and osps == 1");
            return AISX_ERR_INVALID;
        }
        fn = ffs[li];
        big = &big_ff[li];
    }
    if (!*big) {
        AISX_HIPCHK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        *big = true;
    }
    // a workgroup = msk_waves(lpw) waves with lpw channels each
    hipLaunchKernelGGL(fn, dim3(nwg), dim3(64 * msk_waves(p.lpw)), lds, st, p);
    AISX_HIPCHK(hipGetLastError());
    return AISX_OK;
}

// ---------------------------------------------------------------------------
// msk_timing_recovery_cc
// ---------------------------------------------------------------------------
struct aisx_msk {
    int nchan = 0, max_items = 0, out_cap = 0, osps = 1;
    // measurement hook (aisx_msk_set_profiling): hipEvents around the recovery kernel of every stream call
    int prof = 0;
    static constexpr int NEV = 64;
    hipEvent_t pev0[NEV] = {}, pev1[NEV] = {};
    long ncalls_prof = 0;
    int lpw = 64; // channels per wave of the timing-recovery kernel
    int inline_tags = 1; // (AISX_MSK_INLINE_TAGS=0: every tag reset through the general steps, for A/B runs)
    float d_sps = 0, gain = 0, gain_omega = 0, limit = 0;
    static constexpr int carry_cap = MSK_CARRY_MAX, ctag_cap = 64;
    float *d_mu = nullptr, *d_omega = nullptr;
    int* d_div = nullptr;
    cf *d_dly1 = nullptr, *d_dly2 = nullptr, *d_diff1 = nullptr;
    // bit tail state (previous symbol, previous sliced bit): read from [tcur], written to [tcur ^ 1]
    cf* d_tprev[2] = { nullptr, nullptr };
    unsigned char* d_tbit[2] = { nullptr, nullptr };
    int tcur = 0;
    // symbols for the bit tail when the caller takes bits only; two, alternating, so that the
    // bit tail of call k may still read one while call k+1 writes the other (tail stream)
    cf* d_symscratch[2] = { nullptr, nullptr };
    size_t symscratch_len[2] = { 0, 0 };
    int callpar = 0;
    // optional: the bit tail on a stream of its own (aisx_msk_set_tail_stream)
    bool tail_on = false;
    hipStream_t tail_stream = nullptr;
    hipEvent_t ev_msk = nullptr, ev_tail[2] = { nullptr, nullptr };
    unsigned head_start_ticks = 0; // aisx_msk_set_head_start
    hipEvent_t ev_prep = nullptr; // behind the tag prepass of the last aisx_msk_process_stream (aisx_msk_wait_prepass)
    bool ev_prep_set = false;
    bool ev_tail_set[2] = { false, false };
    int* d_produced2 = nullptr; // second internal `produced` array (alternates with d_produced)
    unsigned long long* d_nread = nullptr;
    cf* d_carry[2] = { nullptr, nullptr };
    int* d_carry_len[2] = { nullptr, nullptr };
    tag_rec* d_ctag[2] = { nullptr, nullptr };
    int* d_ctag_n[2] = { nullptr, nullptr };
    msk_ctag* d_ct = nullptr; // this call's time_est tags, compacted (k_msk_tagprep)
    int* d_ct_n = nullptr;
    int ct_cap = 0;
    int cur = 0;
    int *d_produced = nullptr, *d_consumed = nullptr, *d_status = nullptr;
    float *d_mmse = nullptr, *d_atan = nullptr;
    // time-parallel path (k_mskp.h)
    int tp_smax = 0;       // the time-parallel recovery (k_mskp.h): restart points per channel at most; 0 = off, the serial kernel alone
    int tp_min_gap = 64;   // items between restart points at least
    int tp_jw = 16;        // channels per wave of the join kernel
    int tp_join = 1;       // the join: 1 = the serial kernel with fast-forward (k_msk.h, MskParams::ff), 0 = k_mskp_join
    int tp_max_span = 4096; // no unit from a restart point further than this from the next one (tp_join = 1: the serial kernel is faster there)
    int* d_ct_nc = nullptr;
    // the units run on a stream of their own, one call ahead of the join (which needs the previous call's
    // state): everything the prepass and the units leave for the join exists twice, by the call's parity
    hipStream_t s_units = nullptr;
    hipEvent_t ev_entry = nullptr, ev_units[2] = { nullptr, nullptr }, ev_join[2] = { nullptr, nullptr };
    bool ev_join_set[2] = { false, false };
    int max_noutput = 0;   // set_max_noutput_items(): output items one general_work call is offered at most (0: what fits)
    unsigned long long total_in = 0; // items handed to the block so far = absolute offset of the next row's item 0
    msk_ctag* d_ctl = nullptr;
    int* d_ctl_n = nullptr;
    int ctl_cap = 0;
    int* d_nrst = nullptr;
    mskp_rst* d_rst = nullptr;
    mskp_res* d_res = nullptr;
    cf* d_stage[2] = { nullptr, nullptr };
    long stage_stride = 0;
    int* d_ucount = nullptr; // units per length class
    int* d_ulist = nullptr;  // ... and which
    mskp_piece* d_pieces[2] = { nullptr, nullptr };
    int* d_npieces[2] = { nullptr, nullptr };
    long tp_calls = 0;
    // GNU Radio path staging
    cf *d_st_in = nullptr, *d_st_sym = nullptr; // (d_st_sym = d_st_blk + 2: the symbols behind their 16-byte header)
    cf* d_st_blk = nullptr;
    std::vector<cf> st_host; // where header + symbols land on the host
    float *d_st_err = nullptr, *d_st_mu = nullptr;
    unsigned char* d_st_bits = nullptr;
    tag_rec* d_st_tags = nullptr;
    int* d_st_tagn = nullptr;
    int st_in_cap = 0, st_out_cap = 0, st_tag_cap = 0;
};

// What the kernel's LDS rings and the carry buffer are sized for (k_msk.h): one general_work call
// of a single output must fit the carry (forecast(1) + the pre-item), a pair of iterations must
// stay well inside a 64-sample chunk, and omega must stay positive under the clip of :182
// (omega in [d_sps - |limit|, d_sps + |limit|], `limit` is absolute).
static int msk_check_geometry(float d_sps, float gain, float limit)
{
    const float wmin = d_sps - fabsf(limit), wmax = d_sps + fabsf(limit);
    if (!(wmin >= 0.5f) || msk_forecast(d_sps, 1) + 1 > aisx_msk::carry_cap || !(2.f * wmax + 3.f * fabsf(gain) <= 32.f)) {
        set_err("msk_timing_recovery: sps/2 = %g with limit %g and gain %g is outside what the gfx950 kernel is sized for "
                "(sps/2 - |limit| >= 0.5, forecast(1) < %d items, 2 (sps/2 + |limit|) + 3 |gain| <= 32)",
                d_sps, limit, gain, aisx_msk::carry_cap);
        return AISX_ERR_INVALID;
    }
    return AISX_OK;
}
// items per channel one call can produce: without tags every output consumes at least
// 2 (d_sps - |limit|) input items (osps = 1; half of that for osps = 2).  A time_est tag sets
// d_div = 0 (:159): the iteration it resets emits a symbol whatever came before, and it may step
// iidx back by one (:151-154) -- up to two more outputs per tag.  Room for max_items / 64 tags
// per call is added (the stock chain produces one per ~600 samples; a burst gives 3-4 on
// consecutive pairs); a call that needs more ends with AISX_MSK_ST_OUT_FULL.
static int msk_out_cap(const aisx_msk* h)
{
    const double wmin = (double)h->d_sps - fabs((double)h->limit);
    const int tag_room = 2 * std::max(16, h->max_items / 64);
    return ((int)ceil((h->max_items + aisx_msk::carry_cap) / (2.0 * wmin)) + tag_room) * h->osps + 16;
}

static int msk_init_state(aisx_msk* h)
{
    const int nc = h->nchan;
    std::vector<float> mu(nc, 0.5f), om(nc, h->d_sps); // impl :49-56, :71
    AISX_HIPCHK(hipMemcpy(h->d_mu, mu.data(), sizeof(float) * nc, hipMemcpyHostToDevice));
    AISX_HIPCHK(hipMemcpy(h->d_omega, om.data(), sizeof(float) * nc, hipMemcpyHostToDevice));
    AISX_HIPCHK(hipMemset(h->d_div, 0, sizeof(int) * nc));
    AISX_HIPCHK(hipMemset(h->d_dly1, 0, sizeof(cf) * nc));
    AISX_HIPCHK(hipMemset(h->d_dly2, 0, sizeof(cf) * nc));
    AISX_HIPCHK(hipMemset(h->d_diff1, 0, sizeof(cf) * nc));
    for (int k = 0; k < 2; k++) {
        AISX_HIPCHK(hipMemset(h->d_tprev[k], 0, sizeof(cf) * nc));
        AISX_HIPCHK(hipMemset(h->d_tbit[k], 0, nc));
    }
    h->tcur = 0;
    AISX_HIPCHK(hipMemset(h->d_nread, 0, sizeof(unsigned long long) * nc));
    for (int k = 0; k < 2; k++) {
        AISX_HIPCHK(hipMemset(h->d_carry[k], 0, sizeof(cf) * (size_t)nc * aisx_msk::carry_cap));
        AISX_HIPCHK(hipMemset(h->d_carry_len[k], 0, sizeof(int) * nc));
        AISX_HIPCHK(hipMemset(h->d_ctag_n[k], 0, sizeof(int) * nc));
    }
    h->cur = 0;
    h->total_in = 0;
    return AISX_OK;
}

extern "C" int aisx_msk_create(aisx_msk** out, float sps, float gain, float limit, int osps, int nchan, int max_items)
{
    if (!out)
        return AISX_ERR_INVALID;
    *out = nullptr;
    if (nchan < 1 || max_items < 1 || !(sps > 0)) {
        set_err("aisx_msk_create: bad argument");
        return AISX_ERR_INVALID;
    }
    if (!(gain > 0)) { // impl :82
        set_err("Gain must be positive");
        return AISX_ERR_OUT_OF_RANGE;
    }
    if (osps != 1 && osps != 2) { // impl :61
        set_err("osps must be 1 or 2");
        return AISX_ERR_OUT_OF_RANGE;
    }
    int rc = require_device();
    if (rc != AISX_OK)
        return rc;
    if ((rc = msk_check_geometry(msk_setup(sps, gain).d_sps, gain, limit)) != AISX_OK)
        return rc;
    aisx_msk* h = new aisx_msk();
    h->nchan = nchan;
    h->max_items = max_items;
    h->osps = osps;
    h->limit = limit;
    h->d_sps = msk_setup(sps, gain).d_sps; // :70
    h->gain = gain;
    h->gain_omega = msk_setup(sps, gain).gain_omega; // :83
    h->out_cap = msk_out_cap(h);
    {
        // 8 channels per wave, four waves (one per SIMD), 32 channels and 90 KB of LDS per
        // workgroup: fewer lanes per wave = fewer events of other lanes to wait for (a tag costs
        // the whole wave a general pass), and half of each CU's LDS stays free for the stages
        // that run beside this kernel on the other stream.  Measured on the whole flowgraph:
        // 7.2 / 6.3 / 5.5 ms per launch at 16 / 8 / 4 channels per wave; 8 gives the shortest
        // step (at 4 the kernel sits on all 256 CUs and slows the bandwidth-bound stages more)
        h->lpw = 8;
        if (const char* e = exp_env("AISX_MSK_INLINE_TAGS"))
            h->inline_tags = atoi(e) != 0;
        if (const char* e = exp_env("AISX_MSK_TIME_PARALLEL")) // (experiments; the API is aisx_msk_set_time_parallel)
            h->tp_smax = atoi(e) != 0 ? MSKP_SMAX : 0;
        if (const char* e = exp_env("AISX_MSK_TP_SMAX")) // restart points per channel (0: the serial kernel)
            h->tp_smax = std::max(0, std::min(atoi(e), (int)MSKP_SMAX));
        if (const char* e = exp_env("AISX_MSK_TP_GAP"))
            h->tp_min_gap = std::max(0, atoi(e));
        if (const char* e = exp_env("AISX_MSK_TP_JOIN"))
            h->tp_join = atoi(e) != 0;
        if (const char* e = exp_env("AISX_MSK_TP_MAXSPAN"))
            h->tp_max_span = std::max(64, atoi(e));
        if (const char* e = exp_env("AISX_MSK_JW"))
            h->tp_jw = std::max(1, std::min(64, atoi(e)));
        if (const char* e = exp_env("AISX_MSK_MAX_NOUTPUT")) // (experiments; the API is aisx_msk_set_max_noutput_items)
            h->max_noutput = std::max(0, atoi(e));
        if (const char* e = exp_env("AISX_MSK_LPW")) { // (experiments)
            const int v = atoi(e);
            if (v == 4 || v == 8 || v == 16 || v == 32 || v == 64)
                h->lpw = v;
        }
    }
#define CK(e)               \
    do {                    \
        rc = (e);           \
        if (rc != AISX_OK) { \
            aisx_msk_destroy(h); \
            return rc;      \
        }                   \
    } while (0)
    CK(dev_alloc(&h->d_mu, nchan));
    CK(dev_alloc(&h->d_omega, nchan));
    CK(dev_alloc(&h->d_div, nchan));
    CK(dev_alloc(&h->d_dly1, nchan));
    CK(dev_alloc(&h->d_dly2, nchan));
    CK(dev_alloc(&h->d_diff1, nchan));
    for (int k = 0; k < 2; k++) {
        CK(dev_alloc(&h->d_tprev[k], nchan));
        CK(dev_alloc(&h->d_tbit[k], nchan));
    }
    CK(dev_alloc(&h->d_nread, nchan));
    for (int k = 0; k < 2; k++) {
        CK(dev_alloc(&h->d_carry[k], (size_t)nchan * aisx_msk::carry_cap));
        CK(dev_alloc(&h->d_carry_len[k], nchan));
        CK(dev_alloc(&h->d_ctag[k], (size_t)nchan * aisx_msk::ctag_cap));
        CK(dev_alloc(&h->d_ctag_n[k], nchan));
    }
    CK(dev_alloc(&h->d_produced, nchan));
    CK(dev_alloc(&h->d_produced2, nchan));
    CK(dev_alloc(&h->d_consumed, nchan));
    CK(dev_alloc(&h->d_status, nchan));
    CK(dev_alloc(&h->d_mmse, 129 * 8));
    CK(dev_alloc(&h->d_atan, 257));
    if (hipMemcpy(h->d_mmse, aisx_mmse_taps, sizeof(float) * 129 * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->d_atan, aisx_atan_table, sizeof(float) * 257, hipMemcpyHostToDevice) != hipSuccess) {
        set_err("aisx_msk_create: table upload failed");
        aisx_msk_destroy(h);
        return AISX_ERR_HIP;
    }
    CK(msk_init_state(h));
    // the compacted tag list for the usual hand-over capacity; grown on demand
    h->ct_cap = aisx_msk::ctag_cap + 1024;
    CK(dev_alloc(&h->d_ct, (size_t)nchan * (size_t)h->ct_cap));
    CK(dev_alloc(&h->d_ct_n, nchan));
    if (hipDeviceSynchronize() != hipSuccess) {
        aisx_msk_destroy(h);
        return AISX_ERR_HIP;
    }
#undef CK
    *out = h;
    return AISX_OK;
}

// everything the time-parallel path allocates on first use (and nothing else): after this the handle is as if
// the path had never run
static void msk_tp_free(aisx_msk* h)
{
    if (h->s_units) {
        (void)hipStreamSynchronize(h->s_units);
        (void)hipStreamDestroy(h->s_units);
        h->s_units = nullptr;
    }
    auto drop = [](auto*& p) {
        dev_free(p);
        p = nullptr;
    };
    drop(h->d_ctl);
    h->ctl_cap = 0;
    drop(h->d_ctl_n);
    drop(h->d_nrst);
    drop(h->d_rst);
    drop(h->d_res);
    drop(h->d_ucount);
    drop(h->d_ulist);
    drop(h->d_ct_nc);
    if (h->ev_entry)
        (void)hipEventDestroy(h->ev_entry);
    h->ev_entry = nullptr;
    for (int k = 0; k < 2; k++) {
        if (h->ev_units[k])
            (void)hipEventDestroy(h->ev_units[k]);
        if (h->ev_join[k])
            (void)hipEventDestroy(h->ev_join[k]);
        h->ev_units[k] = h->ev_join[k] = nullptr;
        h->ev_join_set[k] = false;
        drop(h->d_stage[k]);
        drop(h->d_pieces[k]);
        drop(h->d_npieces[k]);
    }
}

extern "C" int aisx_msk_destroy(aisx_msk* h)
{
    if (!h)
        return AISX_OK;
    dev_free(h->d_mu);
    dev_free(h->d_omega);
    dev_free(h->d_div);
    dev_free(h->d_dly1);
    dev_free(h->d_dly2);
    dev_free(h->d_diff1);
    for (int k = 0; k < 2; k++) {
        dev_free(h->d_tprev[k]);
        dev_free(h->d_tbit[k]);
    }
    dev_free(h->d_symscratch[0]);
    dev_free(h->d_symscratch[1]);
    dev_free(h->d_produced2);
    if (h->ev_msk)
        (void)hipEventDestroy(h->ev_msk);
    for (int k = 0; k < 2; k++)
        if (h->ev_tail[k])
            (void)hipEventDestroy(h->ev_tail[k]);
    if (h->ev_prep)
        (void)hipEventDestroy(h->ev_prep);
    for (int k = 0; k < aisx_msk::NEV; k++) {
        if (h->pev0[k])
            (void)hipEventDestroy(h->pev0[k]);
        if (h->pev1[k])
            (void)hipEventDestroy(h->pev1[k]);
    }
    dev_free(h->d_ct);
    dev_free(h->d_ct_n);
    msk_tp_free(h);
    dev_free(h->d_nread);
    for (int k = 0; k < 2; k++) {
        dev_free(h->d_carry[k]);
        dev_free(h->d_carry_len[k]);
        dev_free(h->d_ctag[k]);
        dev_free(h->d_ctag_n[k]);
    }
    dev_free(h->d_produced);
    dev_free(h->d_consumed);
    dev_free(h->d_status);
    dev_free(h->d_mmse);
    dev_free(h->d_atan);
    dev_free(h->d_st_in);
    dev_free(h->d_st_blk);
    dev_free(h->d_st_err);
    dev_free(h->d_st_mu);
    dev_free(h->d_st_bits);
    dev_free(h->d_st_tags);
    dev_free(h->d_st_tagn);
    delete h;
    return AISX_OK;
}

extern "C" int aisx_msk_set_gain(aisx_msk* h, float gain)
{
    if (!h)
        return AISX_ERR_INVALID;
    if (!(gain > 0)) {
        h->gain = gain; // the reference stores first, then throws (:81-82): get_gain() shows the bad value
        set_err("Gain must be positive");
        return AISX_ERR_OUT_OF_RANGE;
    }
    // a gain the kernel's rings are not sized for is refused BEFORE anything is stored (as
    // set_limit / set_sps do): later calls keep running with the previous, valid loop gains
    const int rc = msk_check_geometry(h->d_sps, gain, h->limit);
    if (rc != AISX_OK)
        return rc;
    h->gain = gain;
    h->gain_omega = (float)(gain * gain * 0.25);
    return AISX_OK;
}
extern "C" float aisx_msk_get_gain(const aisx_msk* h) { return h ? h->gain : 0.f; }
extern "C" int aisx_msk_set_limit(aisx_msk* h, float limit)
{
    if (!h)
        return AISX_ERR_INVALID;
    const int rc = msk_check_geometry(h->d_sps, h->gain, limit);
    if (rc != AISX_OK)
        return rc; // (the reference accepts any value, :90-92; this build is sized, see msk_check_geometry)
    h->limit = limit;
    h->out_cap = msk_out_cap(h); // callers size their outputs by aisx_msk_out_capacity(): ask again
    return AISX_OK;
}
extern "C" float aisx_msk_get_limit(const aisx_msk* h) { return h ? h->limit : 0.f; }
extern "C" int aisx_msk_set_sps(aisx_msk* h, float sps)
{
    if (!h)
        return AISX_ERR_INVALID;
    const int rc = msk_check_geometry((float)(sps / 2.0), h->gain, h->limit);
    if (rc != AISX_OK)
        return rc;
    h->d_sps = (float)(sps / 2.0); // :70
    h->out_cap = msk_out_cap(h);
    std::vector<float> om(h->nchan, h->d_sps); // :71 d_omega = d_sps
    AISX_HIPCHK(hipMemcpy(h->d_omega, om.data(), sizeof(float) * h->nchan, hipMemcpyHostToDevice));
    return AISX_OK;
}
extern "C" float aisx_msk_get_sps(const aisx_msk* h) { return h ? h->d_sps : 0.f; }
extern "C" int aisx_msk_forecast(const aisx_msk* h, int noutput_items)
{
    return h ? msk_forecast(h->d_sps, noutput_items) : AISX_ERR_INVALID;
}
extern "C" int aisx_msk_set_max_noutput_items(aisx_msk* h, int max_noutput_items)
{
    if (!h || max_noutput_items < 0)
        return AISX_ERR_INVALID;
    h->max_noutput = max_noutput_items;
    return AISX_OK;
}
extern "C" int aisx_msk_set_time_parallel(aisx_msk* h, int restart_points_per_channel, int join_kernel, int max_unit_items)
{
    if (!h || restart_points_per_channel < 0)
        return AISX_ERR_INVALID;
    h->tp_smax = std::min(restart_points_per_channel, (int)MSKP_SMAX);
    if (join_kernel >= 0)
        h->tp_join = join_kernel != 0;
    if (max_unit_items > 0)
        h->tp_max_span = std::max(64, max_unit_items);
    return AISX_OK;
}
extern "C" int aisx_msk_get_max_noutput_items(const aisx_msk* h) { return h ? h->max_noutput : AISX_ERR_INVALID; }
extern "C" int aisx_msk_out_capacity(const aisx_msk* h) { return h ? h->out_cap : AISX_ERR_INVALID; }
extern "C" int aisx_msk_reset(aisx_msk* h)
{
    if (!h)
        return AISX_ERR_INVALID;
    const int rc = msk_init_state(h);
    if (rc != AISX_OK)
        return rc;
    AISX_HIPCHK(hipDeviceSynchronize()); // (null-stream fills vs. the caller's non-blocking streams)
    return AISX_OK;
}

static void msk_fill_common(aisx_msk* h, MskParams& p)
{
    p.nchan = h->nchan;
    p.d_sps = h->d_sps;
    p.gain = h->gain;
    p.gain_omega = h->gain_omega;
    p.limit = h->limit;
    p.osps = h->osps;
    p.mu = h->d_mu;
    p.omega = h->d_omega;
    p.div = h->d_div;
    p.dly1 = h->d_dly1;
    p.dly2 = h->d_dly2;
    p.diff1 = h->d_diff1;
    p.nread = h->d_nread;
    p.carry_in = h->d_carry[h->cur];
    p.carry_out = h->d_carry[h->cur ^ 1];
    p.carry_len_in = h->d_carry_len[h->cur];
    p.carry_len_out = h->d_carry_len[h->cur ^ 1];
    p.carry_cap = aisx_msk::carry_cap;
    p.ctag_out = h->d_ctag[h->cur ^ 1];
    p.ctag_n_out = h->d_ctag_n[h->cur ^ 1];
    p.ctag_cap = aisx_msk::ctag_cap;
    p.ct = h->d_ct;
    p.ct_n = h->d_ct_n;
    p.ct_cap = h->ct_cap;
    p.consumed = h->d_consumed;
    p.status = h->d_status;
    p.mmse = h->d_mmse;
    p.lds_tab_off = msk_lds_taboff(h->lpw);
    p.lpw = h->lpw;
    p.lds_wave_stride = msk_lds_wave(h->lpw);
    p.tq_stride = h->lpw;
    p.tq_private = 0;
    p.lds_ring_off = msk_lds_ringoff(h->lpw);
    p.inline_tags = h->inline_tags;
    p.max_noutput = h->max_noutput;
    p.ff = 0;
    p.nrst = nullptr;
    p.rst = nullptr;
    p.res = nullptr;
    p.pieces = nullptr;
    p.npieces = nullptr;
    p.ct_nc = nullptr;
}

// compacts (carried tags + this call's tags) into h->d_ct for the kernel launch that follows
static int msk_launch_tagprep(aisx_msk* h, const tag_rec* d_tags, const int* d_tag_counts, int tag_cap, hipStream_t st, int* d_ct_nc = nullptr)
{
    const int need = aisx_msk::ctag_cap + (d_tags ? tag_cap : 0);
    int rc;
    if (need > h->ct_cap || !h->d_ct) {
        AISX_HIPCHK(hipStreamSynchronize(st));
        dev_free(h->d_ct);
        h->d_ct = nullptr;
        h->ct_cap = 0;
        if ((rc = dev_alloc(&h->d_ct, (size_t)h->nchan * (size_t)need)) != AISX_OK)
            return rc;
        h->ct_cap = need;
        if (!h->d_ct_n && (rc = dev_alloc(&h->d_ct_n, h->nchan)) != AISX_OK)
            return rc;
        // dev_alloc's zero fill runs on the null stream: it must not trail into the kernels on `st`
        AISX_HIPCHK(hipDeviceSynchronize());
    }
    TagPrepParams t;
    t.nchan = h->nchan;
    t.ctag_in = h->d_ctag[h->cur];
    t.ctag_n_in = h->d_ctag_n[h->cur];
    t.ctag_cap = aisx_msk::ctag_cap;
    t.tags = d_tags;
    t.tag_count = d_tag_counts;
    t.tag_cap = tag_cap;
    t.nread = h->d_nread;
    t.ct = h->d_ct;
    t.ct_n = h->d_ct_n;
    t.ct_cap = h->ct_cap;
    t.ct_nc = d_ct_nc;
    t.ctl_new = nullptr;
    t.ctl_new_n = nullptr;
    t.ctl_new_cap = 0;
    t.ctl_new_pre = 0;
    t.W = 0;
    hipLaunchKernelGGL(k_msk_tagprep, dim3((h->nchan + 3) / 4), dim3(256), 0, st, t); // a wave per channel
    AISX_HIPCHK(hipGetLastError());
    return AISX_OK;
}

// the NRZI bit tail over the symbols the timing-recovery kernel just wrote
static int msk_launch_bittail(aisx_msk* h, const cf* syms, long sym_stride, const int* produced, uint8_t* bits,
                              long bit_stride, int max_out, hipStream_t st)
{
    BitTailParams b;
    b.nchan = h->nchan;
    b.syms = syms;
    b.sym_stride = sym_stride;
    b.produced = produced;
    b.bits = bits;
    b.bit_stride = bit_stride;
    b.prev_sym_in = h->d_tprev[h->tcur];
    b.prev_bit_in = h->d_tbit[h->tcur];
    b.prev_sym_out = h->d_tprev[h->tcur ^ 1];
    b.prev_bit_out = h->d_tbit[h->tcur ^ 1];
    b.atan_tab = h->d_atan;
    const int nseg = std::max(1, (max_out + BT_SEG - 1) / BT_SEG);
    hipLaunchKernelGGL(k_bittail, dim3(nseg, h->nchan), dim3(BT_T), 0, st, b);
    AISX_HIPCHK(hipGetLastError());
    h->tcur ^= 1;
    return AISX_OK;
}

// ---- the time-parallel path -------------------------------------------------------------------
static bool msk_tp_applies(const aisx_msk* h, const float* d_err, const float* d_mu)
{
    // (osps = 2 and the err / mu ports stay with the serial kernel: after a restart the first err
    // of a unit would need the previous unit's last nlin_out)
    return h->tp_smax > 0 && h->osps == 1 && !d_err && !d_mu &&
           mskp_geometry_ok(h->d_sps, h->gain, h->limit, h->max_items + aisx_msk::carry_cap);
}

static int msk_tp_buffers(aisx_msk* h, int tag_cap, hipStream_t st)
{
    int rc;
    const int need = MSKP_TPRE + tag_cap + 1;
    bool fresh = false;
    if (need > h->ctl_cap || !h->d_ctl) {
        AISX_HIPCHK(hipStreamSynchronize(st));
        if (h->s_units)
            AISX_HIPCHK(hipStreamSynchronize(h->s_units));
        dev_free(h->d_ctl);
        h->d_ctl = nullptr;
        h->ctl_cap = 0;
        if ((rc = dev_alloc(&h->d_ctl, 2 * (size_t)h->nchan * (size_t)need)) != AISX_OK)
            return rc;
        h->ctl_cap = need;
        fresh = true;
    }
    if (!h->d_rst) {
        // (all or nothing: a failed allocation half way must not leave a handle that looks set up)
        struct Undo {
            aisx_msk* h;
            bool armed = true;
            ~Undo()
            {
                if (armed)
                    msk_tp_free(h);
            }
        } undo{ h };
        const size_t nc = (size_t)h->nchan;
        h->stage_stride = mskp_stage_stride(h->max_items + aisx_msk::carry_cap, h->d_sps, h->gain, h->limit);
        if ((rc = dev_alloc(&h->d_ctl_n, 2 * nc)) != AISX_OK || (rc = dev_alloc(&h->d_nrst, 2 * nc)) != AISX_OK ||
            (rc = dev_alloc(&h->d_rst, 2 * nc * MSKP_SMAX)) != AISX_OK || (rc = dev_alloc(&h->d_res, 2 * nc * MSKP_SMAX)) != AISX_OK ||
            (rc = dev_alloc(&h->d_ucount, 16)) != AISX_OK || (rc = dev_alloc(&h->d_ct_nc, nc)) != AISX_OK ||
            (rc = dev_alloc(&h->d_ulist, 2 * nc * MSKP_SMAX * MSKP_NCLS)) != AISX_OK)
            return rc;
        AISX_HIPCHK(hipStreamCreateWithFlags(&h->s_units, hipStreamNonBlocking));
        AISX_HIPCHK(hipEventCreateWithFlags(&h->ev_entry, hipEventDisableTiming));
        for (int k = 0; k < 2; k++) {
            AISX_HIPCHK(hipEventCreateWithFlags(&h->ev_units[k], hipEventDisableTiming));
            AISX_HIPCHK(hipEventCreateWithFlags(&h->ev_join[k], hipEventDisableTiming));
        }
        for (int k = 0; k < 2; k++)
            if ((rc = dev_alloc(&h->d_stage[k], nc * (size_t)h->stage_stride)) != AISX_OK ||
                (rc = dev_alloc(&h->d_pieces[k], nc * MSKP_SMAX)) != AISX_OK || (rc = dev_alloc(&h->d_npieces[k], nc)) != AISX_OK)
                return rc;
        undo.armed = false;
        fresh = true;
    }
    if (fresh) // dev_alloc's zero fill runs on the null stream: it must not trail into the kernels on `st`
        AISX_HIPCHK(hipDeviceSynchronize());
    return AISX_OK;
}

static int msk_process_stream(aisx_msk* h, const aisx_cf32* d_in, long in_stride, int n, const aisx_tag* d_tags,
                              const int* d_tag_counts, int tag_cap, aisx_cf32* d_syms, float* d_err, float* d_mu, uint8_t* d_bits,
                              long out_stride, int* d_produced, void* stream, void* ready_event)
{
    if (!h || !d_in || n < 1 || n > h->max_items || in_stride < n || (d_tags && (!d_tag_counts || tag_cap < 1))) {
        set_err("aisx_msk_process_stream: bad argument");
        return AISX_ERR_INVALID;
    }
    if (out_stride < 1) {
        set_err("aisx_msk_process_stream: out_stride < 1");
        return AISX_ERR_INVALID;
    }
    if (out_stride >= (1L << 23)) {
        set_err("aisx_msk_process_stream: out_stride %ld too large (the 64 rows of a wave must lie within 4 GiB)", out_stride);
        return AISX_ERR_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool tp = msk_tp_applies(h, d_err, d_mu);
    const int par = h->callpar;
    h->callpar ^= 1;
    int rc, t_smax = 0;
    bool tp_sorted = false;
    const size_t nc = (size_t)h->nchan;
    // this call's copies of what the prepass and the units leave for the join
    msk_ctag* ctl = nullptr;
    int *ctl_n = nullptr, *nrst = nullptr, *ucount = nullptr, *ulist = nullptr;
    mskp_rst* rst = nullptr;
    mskp_res* res = nullptr;
    hipStream_t su = st; // where the prepass and the units run
    if (tp) {
        if ((rc = msk_tp_buffers(h, d_tags ? tag_cap : 0, st)) != AISX_OK)
            return rc;
        ctl = h->d_ctl + (size_t)par * nc * (size_t)h->ctl_cap;
        ctl_n = h->d_ctl_n + par * nc;
        nrst = h->d_nrst + par * nc;
        rst = h->d_rst + par * nc * MSKP_SMAX;
        res = h->d_res + par * nc * MSKP_SMAX;
        ucount = h->d_ucount + par * 8;
        ulist = h->d_ulist + par * nc * MSKP_SMAX * MSKP_NCLS;
        // The units need the samples and the tags of this call, nothing of the call before: they run
        // on their own stream, beside the join of the previous call.  They start when the caller says
        // the inputs are there (ready_event; without one: when `stream` gets here), when the join of
        // two calls ago has let go of this parity's records and the bit tail of its staging rows.
        // (experiment switches, read once: units on the call's stream; unsorted unit list)
        static const bool one_stream = exp_env("AISX_MSK_TP_ONE_STREAM") != nullptr;
        if (!one_stream)
            su = h->s_units;
        if (su != st) {
            if (ready_event) {
                AISX_HIPCHK(hipStreamWaitEvent(su, (hipEvent_t)ready_event, 0));
            } else {
                AISX_HIPCHK(hipEventRecord(h->ev_entry, st));
                AISX_HIPCHK(hipStreamWaitEvent(su, h->ev_entry, 0));
            }
            if (h->ev_join_set[par])
                AISX_HIPCHK(hipStreamWaitEvent(su, h->ev_join[par], 0));
        }
        if (h->tail_on && h->ev_tail_set[par])
            AISX_HIPCHK(hipStreamWaitEvent(su, h->ev_tail[par], 0));
        MskpPrepParams t;
        t.nchan = h->nchan;
        t.tags = (const tag_rec*)d_tags;
        t.tag_count = d_tag_counts;
        t.tag_cap = tag_cap;
        t.W = h->total_in;
        t.n = n;
        t.d_sps = h->d_sps;
        t.gain = h->gain;
        t.limit = h->limit;
        t.ctl = ctl;
        t.ctl_n = ctl_n;
        t.ctl_cap = h->ctl_cap;
        // (units run blind to the general_work calls: with a max_noutput_items the call boundaries must
        // leave an un-blocked loop alone, which needs d_sps >= 2 -- see mskp_body's walk)
        t.smax = (h->max_noutput > 0 && h->d_sps < 2.0f) ? 0 : h->tp_smax;
        t_smax = t.smax;
        t.nrst = nrst;
        t.rst = rst;
        t.stage_stride = h->stage_stride;
        t.tail = mskp_tail(h->d_sps);
        t.min_gap = h->tp_min_gap;
        t.max_span = h->tp_join ? h->tp_max_span : 0x3fffffff;
        // units sorted by length need every row within 4 GiB of the first (32-bit buffer offsets)
        static const bool tp_unsorted_env = exp_env("AISX_MSK_TP_UNSORTED") != nullptr;
        tp_sorted = (double)h->nchan * (double)in_stride * 8.0 < 4294000000.0 && !tp_unsorted_env;
        t.ucount = tp_sorted ? ucount : nullptr;
        t.ulist = ulist;
        t.ucap = (long)h->nchan * MSKP_SMAX;
        if (tp_sorted)
            AISX_HIPCHK(hipMemsetAsync(ucount, 0, sizeof(int) * 8, su));
        hipLaunchKernelGGL(k_mskp_prep, dim3(h->nchan), dim3(64), 0, su, t);
        AISX_HIPCHK(hipGetLastError());
    } else if ((rc = msk_launch_tagprep(h, (const tag_rec*)d_tags, d_tag_counts, tag_cap, st)) != AISX_OK) {
        return rc;
    }
    if (h->ev_prep) { // (the caller's tag records have been read)
        AISX_HIPCHK(hipEventRecord(h->ev_prep, su));
        h->ev_prep_set = true;
    }
    cf* syms = (cf*)d_syms;
    if (h->tail_on && h->ev_tail_set[par]) // the bit tail of two calls ago may still read this parity's buffers
        AISX_HIPCHK(hipStreamWaitEvent(st, h->ev_tail[par], 0));
    if (!syms) { // the kernel always writes symbols (the bit tail reads them back): give them a home
        const size_t need = (size_t)h->nchan * (size_t)out_stride;
        if (need > h->symscratch_len[par]) {
            AISX_HIPCHK(hipStreamSynchronize(st));
            dev_free(h->d_symscratch[par]);
            h->d_symscratch[par] = nullptr;
            h->symscratch_len[par] = 0;
            if ((rc = dev_alloc(&h->d_symscratch[par], need)) != AISX_OK)
                return rc;
            AISX_HIPCHK(hipDeviceSynchronize()); // (the zero fill runs on the null stream)
            h->symscratch_len[par] = need;
        }
        syms = h->d_symscratch[par];
    }
    int* produced = d_produced ? d_produced : (par ? h->d_produced2 : h->d_produced);
    const int out_cap = (int)std::min<long>(out_stride, 0x7fffffff);
    if (tp) {
        MskpParams p;
        p.nchan = h->nchan;
        p.d_sps = h->d_sps;
        p.gain = h->gain;
        p.gain_omega = h->gain_omega;
        p.limit = h->limit;
        p.mu = h->d_mu;
        p.omega = h->d_omega;
        p.div = h->d_div;
        p.dly1 = h->d_dly1;
        p.dly2 = h->d_dly2;
        p.diff1 = h->d_diff1;
        p.nread = h->d_nread;
        p.in = (const cf*)d_in;
        p.in_stride = in_stride;
        p.n = n;
        p.carry_in = h->d_carry[h->cur];
        p.carry_out = h->d_carry[h->cur ^ 1];
        p.carry_len_in = h->d_carry_len[h->cur];
        p.carry_len_out = h->d_carry_len[h->cur ^ 1];
        p.carry_cap = aisx_msk::carry_cap;
        p.ctag_in = h->d_ctag[h->cur];
        p.ctag_n_in = h->d_ctag_n[h->cur];
        p.ctag_out = h->d_ctag[h->cur ^ 1];
        p.ctag_n_out = h->d_ctag_n[h->cur ^ 1];
        p.ctag_cap = aisx_msk::ctag_cap;
        p.ctl = ctl;
        p.ctl_n = ctl_n;
        p.ctl_cap = h->ctl_cap;
        p.smax = h->tp_smax;
        p.nrst = nrst;
        p.rst = rst;
        p.res = res;
        p.stage = h->d_stage[par];
        p.stage_stride = h->stage_stride;
        p.syms = syms;
        p.out_stride = out_stride;
        p.out_cap = out_cap;
        p.pieces = h->d_pieces[par];
        p.npieces = h->d_npieces[par];
        p.produced = produced;
        p.consumed = h->d_consumed;
        p.status = h->d_status;
        p.mmse = h->d_mmse;
        p.W = h->total_in;
        p.look = mskp_look(h->d_sps, h->limit);
        p.padv = mskp_padv(h->d_sps, h->gain, h->limit);
        p.padv_inv = mskp_padv_inv(h->d_sps, h->gain, h->limit);
        p.jw = h->tp_jw;
        p.ucount = tp_sorted ? ucount : nullptr;
        p.ulist = ulist;
        p.ucap = (long)h->nchan * MSKP_SMAX;
        p.tail = mskp_tail(h->d_sps);
        p.max_noutput = h->max_noutput;
        static bool attr_set = false;
        if (!attr_set) {
            AISX_HIPCHK(hipFuncSetAttribute((const void*)k_mskp_units, hipFuncAttributeMaxDynamicSharedMemorySize, MSKP_LDS_BYTES));
            AISX_HIPCHK(hipFuncSetAttribute((const void*)k_mskp_join, hipFuncAttributeMaxDynamicSharedMemorySize, MSKP_LDS_BYTES));
            attr_set = true;
        }
        if (t_smax > 0) {
            const long units = (long)h->nchan * h->tp_smax;
            hipLaunchKernelGGL(k_mskp_units, dim3((unsigned)((units + 63) / 64 + (tp_sorted ? MSKP_NCLS : 0))), dim3(64), MSKP_LDS_BYTES, su, p);
            AISX_HIPCHK(hipGetLastError());
        }
        if (su != st) { // the join, on the caller's stream, behind the units
            AISX_HIPCHK(hipEventRecord(h->ev_units[par], su));
            AISX_HIPCHK(hipStreamWaitEvent(st, h->ev_units[par], 0));
        }
        if (h->tp_join) {
            // the serial kernel as the join: the loop from the carried state, fast-forwarded through the units;
            // its tag list = the tags the scheduler still held + this call's, as the prepass compacted them
            const int need = aisx_msk::ctag_cap + (h->ctl_cap - MSKP_TPRE);
            if (need > h->ct_cap || !h->d_ct) {
                AISX_HIPCHK(hipStreamSynchronize(st));
                dev_free(h->d_ct);
                h->d_ct = nullptr;
                h->ct_cap = 0;
                if ((rc = dev_alloc(&h->d_ct, nc * (size_t)need)) != AISX_OK)
                    return rc;
                h->ct_cap = need;
                AISX_HIPCHK(hipDeviceSynchronize());
            }
            TagPrepParams tg;
            tg.nchan = h->nchan;
            tg.ctag_in = h->d_ctag[h->cur];
            tg.ctag_n_in = h->d_ctag_n[h->cur];
            tg.ctag_cap = aisx_msk::ctag_cap;
            tg.tags = nullptr;
            tg.tag_count = nullptr;
            tg.tag_cap = 0;
            tg.nread = h->d_nread;
            tg.ct = h->d_ct;
            tg.ct_n = h->d_ct_n;
            tg.ct_cap = h->ct_cap;
            tg.ct_nc = h->d_ct_nc;
            tg.ctl_new = ctl;
            tg.ctl_new_n = ctl_n;
            tg.ctl_new_cap = h->ctl_cap;
            tg.ctl_new_pre = MSKP_TPRE;
            tg.W = h->total_in;
            hipLaunchKernelGGL(k_msk_tagprep, dim3((h->nchan + 3) / 4), dim3(256), 0, st, tg);
            AISX_HIPCHK(hipGetLastError());
            MskParams m;
            msk_fill_common(h, m);
            m.in = (const cf*)d_in;
            m.in_stride = in_stride;
            m.n = n;
            m.stream_mode = 1;
            m.gr_ninput = 0;
            m.gr_noutput = 0;
            m.syms = syms;
            m.err = nullptr;
            m.mu_out = nullptr;
            m.out_stride = out_stride;
            m.sym_al16 = 0; // (behind a fast-forward a channel's symbol count may be odd)
            m.out_cap = out_cap;
            m.produced = produced;
            m.inline_tags = 0; // (every tag reset through the general step, where the junctions are looked at)
            m.ff = 1;
            m.nrst = nrst;
            m.rst = rst;
            m.res = res;
            m.pieces = h->d_pieces[par];
            m.npieces = h->d_npieces[par];
            m.ct_nc = h->d_ct_nc;
            if ((rc = msk_launch(m, (h->nchan + msk_wg_channels(h->lpw) - 1) / msk_wg_channels(h->lpw), st)) != AISX_OK)
                return rc;
        } else {
            hipLaunchKernelGGL(k_mskp_join, dim3((h->nchan + h->tp_jw - 1) / h->tp_jw), dim3(64), MSKP_LDS_BYTES, st, p);
            AISX_HIPCHK(hipGetLastError());
        }
        if (su != st) {
            AISX_HIPCHK(hipEventRecord(h->ev_join[par], st));
            h->ev_join_set[par] = true;
        }
        h->tp_calls++;
    } else {
        MskParams p;
        msk_fill_common(h, p);
        p.in = (const cf*)d_in;
        p.in_stride = in_stride;
        p.n = n;
        p.stream_mode = 1;
        p.gr_ninput = 0;
        p.gr_noutput = 0;
        p.syms = syms;
        p.err = d_err;
        p.mu_out = d_mu;
        p.out_stride = out_stride;
        p.sym_al16 = ((uintptr_t)syms % 16 == 0) && (out_stride % 2 == 0);
        p.out_cap = out_cap;
        p.produced = produced;
        const int evi = (int)(h->ncalls_prof % aisx_msk::NEV);
        if (h->prof)
            AISX_HIPCHK(hipEventRecord(h->pev0[evi], st));
        if ((rc = msk_launch(p, (h->nchan + msk_wg_channels(h->lpw) - 1) / msk_wg_channels(h->lpw), st)) != AISX_OK)
            return rc;
        if (h->prof) {
            AISX_HIPCHK(hipEventRecord(h->pev1[evi], st));
            h->ncalls_prof++;
        }
    }
    h->cur ^= 1;
    h->total_in += (unsigned long long)n;
    MskpGatherParams g;
    if (tp) {
        g.nchan = h->nchan;
        g.pieces = h->d_pieces[par];
        g.npieces = h->d_npieces[par];
        g.stage = h->d_stage[par];
        g.stage_stride = h->stage_stride;
        g.syms = syms;
        g.out_stride = out_stride;
        if (d_syms || !d_bits || !h->tail_on) { // the caller's own symbol rows are complete when `stream` is
            hipLaunchKernelGGL(k_mskp_gather, dim3(MSKP_GATHER_X, h->nchan), dim3(256), 0, st, g);
            AISX_HIPCHK(hipGetLastError());
            // this gather reads d_stage[par] / d_res[par], which the units of the call after next overwrite on
            // their own stream: they wait for ev_join[par], so it has to stand BEHIND the gather (without a
            // bit tail on another stream nothing else orders the two)
            if (h->ev_join_set[par])
                AISX_HIPCHK(hipEventRecord(h->ev_join[par], st));
        }
    }
    if (d_bits) {
        // a call produces at most forecast^-1(n + carry) symbols; out_cap bounds it too
        const double wmin = (double)h->d_sps - fabs((double)h->limit);
        const int max_out = std::min<long>(out_cap, (long)ceil((n + aisx_msk::carry_cap) / (2.0 * wmin)) * h->osps + 16);
        hipStream_t ts = st;
        if (h->tail_on) { // the bit tail has no part in the recurrence: let the next call start
            AISX_HIPCHK(hipEventRecord(h->ev_msk, st));
            AISX_HIPCHK(hipStreamWaitEvent(h->tail_stream, h->ev_msk, 0));
            ts = h->tail_stream;
            if (tp && !d_syms) { // the units' symbols join the others on the tail stream
                hipLaunchKernelGGL(k_mskp_gather, dim3(MSKP_GATHER_X, h->nchan), dim3(256), 0, ts, g);
                AISX_HIPCHK(hipGetLastError());
            }
        }
        if ((rc = msk_launch_bittail(h, syms, out_stride, produced, d_bits, out_stride, max_out, ts)) != AISX_OK)
            return rc;
        if (h->tail_on) {
            AISX_HIPCHK(hipEventRecord(h->ev_tail[par], h->tail_stream));
            h->ev_tail_set[par] = true;
        }
    }
    return AISX_OK;
}

extern "C" int aisx_msk_process_stream(aisx_msk* h, const aisx_cf32* d_in, long in_stride, int n,
                                       const aisx_tag* d_tags, const int* d_tag_counts, int tag_cap, aisx_cf32* d_syms,
                                       float* d_err, float* d_mu, uint8_t* d_bits, long out_stride, int* d_produced,
                                       void* stream)
{
    return msk_process_stream(h, d_in, in_stride, n, d_tags, d_tag_counts, tag_cap, d_syms, d_err, d_mu, d_bits, out_stride, d_produced,
                              stream, nullptr);
}

extern "C" int aisx_msk_process_stream_after(aisx_msk* h, const aisx_cf32* d_in, long in_stride, int n,
                                             const aisx_tag* d_tags, const int* d_tag_counts, int tag_cap, aisx_cf32* d_syms,
                                             float* d_err, float* d_mu, uint8_t* d_bits, long out_stride, int* d_produced,
                                             void* stream, void* ready_event)
{
    return msk_process_stream(h, d_in, in_stride, n, d_tags, d_tag_counts, tag_cap, d_syms, d_err, d_mu, d_bits, out_stride, d_produced,
                              stream, ready_event);
}

extern "C" int aisx_msk_set_tail_stream(aisx_msk* h, void* tail_stream, int enable)
{
    if (!h)
        return AISX_ERR_INVALID;
    if (!enable) {
        h->tail_on = false;
        return AISX_OK;
    }
    if (!h->ev_msk)
        AISX_HIPCHK(hipEventCreateWithFlags(&h->ev_msk, hipEventDisableTiming));
    for (int k = 0; k < 2; k++)
        if (!h->ev_tail[k])
            AISX_HIPCHK(hipEventCreateWithFlags(&h->ev_tail[k], hipEventDisableTiming));
    h->tail_stream = (hipStream_t)tail_stream;
    h->tail_on = true;
    return AISX_OK;
}

extern "C" int aisx_msk_wait_tail(aisx_msk* h, void* stream)
{
    if (!h)
        return AISX_ERR_INVALID;
    if (h->tail_on)
        for (int k = 0; k < 2; k++)
            if (h->ev_tail_set[k])
                AISX_HIPCHK(hipStreamWaitEvent((hipStream_t)stream, h->ev_tail[k], 0));
    return AISX_OK;
}

extern "C" int aisx_msk_wait_prepass(aisx_msk* h, void* stream)
{
    if (!h)
        return AISX_ERR_INVALID;
    if (!h->ev_prep) { // first use: from now on every aisx_msk_process_stream records the event
        AISX_HIPCHK(hipEventCreateWithFlags(&h->ev_prep, hipEventDisableTiming));
        return AISX_OK;
    }
    if (h->ev_prep_set) {
        AISX_HIPCHK(hipStreamWaitEvent((hipStream_t)stream, h->ev_prep, 0));
        // The event fires when the tag prepass ends -- the same moment the recovery kernel behind
        // it becomes ready on its own stream: which of the two queues the dispatcher serves first
        // is then a race (it used to be decided by two already-satisfied barrier packets that
        // happened to stand behind this one on `stream`: ~10 us).  A wave that sleeps for
        // aisx_msk_set_head_start()'s microseconds on `stream` decides it (off unless asked for).
        if (h->head_start_ticks > 0) {
            hipLaunchKernelGGL(k_msk_headstart, dim3(1), dim3(64), 0, (hipStream_t)stream, h->head_start_ticks);
            AISX_HIPCHK(hipGetLastError());
        }
    }
    return AISX_OK;
}

extern "C" int aisx_msk_set_head_start(aisx_msk* h, int microseconds)
{
    if (!h || microseconds < 0 || microseconds > 1000) {
        set_err("aisx_msk_set_head_start: 0..1000 microseconds");
        return AISX_ERR_INVALID;
    }
    // the sleeping wave counts wall_clock64() ticks: the constant-rate counter, hipDeviceAttributeWallClockRate kHz
    int dev = 0, khz = 0;
    AISX_HIPCHK(hipGetDevice(&dev));
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
        khz = 100000; // gfx950: 100 MHz
    h->head_start_ticks = (unsigned)((long long)microseconds * khz / 1000);
    return AISX_OK;
}

extern "C" int aisx_msk_geometry(const aisx_msk* h, int* nchan, int* max_items)
{
    if (!h)
        return AISX_ERR_INVALID;
    if (nchan)
        *nchan = h->nchan;
    if (max_items)
        *max_items = h->max_items;
    return AISX_OK;
}

extern "C" int aisx_msk_placement(const aisx_msk* h, int* workgroups, int* lds_bytes_per_workgroup)
{
    if (!h)
        return AISX_ERR_INVALID;
    if (workgroups)
        *workgroups = (h->nchan + msk_wg_channels(h->lpw) - 1) / msk_wg_channels(h->lpw);
    if (lds_bytes_per_workgroup)
        *lds_bytes_per_workgroup = msk_lds_bytes(h->lpw);
    return AISX_OK;
}

// what the time-parallel path made of the last call (diagnostics; waits for `stream`)
extern "C" int aisx_msk_set_profiling(aisx_msk* h, int on)
{
    if (!h)
        return AISX_ERR_INVALID;
    if (on && !h->pev0[0]) {
        // all events or none: created into temporaries, committed to the handle once every one exists
        hipEvent_t e0[aisx_msk::NEV] = {}, e1[aisx_msk::NEV] = {};
        bool ok = true;
        for (int k = 0; k < aisx_msk::NEV && ok; k++)
            ok = hipEventCreate(&e0[k]) == hipSuccess && hipEventCreate(&e1[k]) == hipSuccess;
        if (!ok) {
            for (int k = 0; k < aisx_msk::NEV; k++) {
                if (e0[k])
                    (void)hipEventDestroy(e0[k]);
                if (e1[k])
                    (void)hipEventDestroy(e1[k]);
            }
            set_err("aisx_msk_set_profiling: hipEventCreate failed");
            return AISX_ERR_HIP;
        }
        for (int k = 0; k < aisx_msk::NEV; k++) {
            h->pev0[k] = e0[k];
            h->pev1[k] = e1[k];
        }
    }
    h->prof = on ? 1 : 0;
    h->ncalls_prof = 0;
    return AISX_OK;
}

extern "C" int aisx_msk_kernel_ms_history(aisx_msk* h, float* ms, int cap, int* n)
{
    if (!h || !ms || !n || !h->pev0[0])
        return AISX_ERR_INVALID;
    const long have = h->ncalls_prof < aisx_msk::NEV ? h->ncalls_prof : aisx_msk::NEV;
    int w = 0;
    for (long k = h->ncalls_prof - have; k < h->ncalls_prof && w < cap; k++) {
        const int evi = (int)(k % aisx_msk::NEV);
        AISX_HIPCHK(hipEventSynchronize(h->pev1[evi]));
        AISX_HIPCHK(hipEventElapsedTime(&ms[w], h->pev0[evi], h->pev1[evi]));
        w++;
    }
    *n = w;
    return AISX_OK;
}

extern "C" int aisx_msk_restart_stats(aisx_msk* h, long long* out10, void* stream)
{
    if (!h || !out10)
        return AISX_ERR_INVALID;
    long long* const out6 = out10; // (ten entries: include/aisx.h)
    for (int i = 0; i < 10; i++)
        out6[i] = 0;
    out6[5] = h->tp_calls;
    if (!h->d_rst || h->tp_calls == 0)
        return AISX_OK;
    const int par = h->callpar ^ 1; // the call before this one
    const size_t nc = (size_t)h->nchan;
    std::vector<int> nrst(nc), np(nc);
    std::vector<mskp_piece> pc(nc * MSKP_SMAX);
    std::vector<mskp_res> rs(nc * MSKP_SMAX);
    std::vector<mskp_rst> rp(nc * MSKP_SMAX);
    AISX_HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    // (a pipelined caller's join runs on a stream of its own: its records are complete behind ev_join)
    if (h->ev_join_set[par])
        AISX_HIPCHK(hipEventSynchronize(h->ev_join[par]));
    AISX_HIPCHK(hipMemcpy(nrst.data(), h->d_nrst + par * nc, sizeof(int) * nc, hipMemcpyDeviceToHost));
    AISX_HIPCHK(hipMemcpy(np.data(), h->d_npieces[par], sizeof(int) * nc, hipMemcpyDeviceToHost));
    AISX_HIPCHK(hipMemcpy(pc.data(), h->d_pieces[par], sizeof(mskp_piece) * pc.size(), hipMemcpyDeviceToHost));
    AISX_HIPCHK(hipMemcpy(rs.data(), h->d_res + par * nc * MSKP_SMAX, sizeof(mskp_res) * rs.size(), hipMemcpyDeviceToHost));
    AISX_HIPCHK(hipMemcpy(rp.data(), h->d_rst + par * nc * MSKP_SMAX, sizeof(mskp_rst) * rp.size(), hipMemcpyDeviceToHost));
    for (size_t c = 0; c < nc; c++) {
        out6[0] += nrst[c];                     // restart points chosen
        out6[1] += np[c];                       // units whose run was taken over
        for (int i = 0; i < np[c]; i++)
            out6[2] += pc[c * MSKP_SMAX + i].cnt; // symbols that came from units
        for (int i = 0; i < nrst[c]; i++) {
            out6[3] += rs[c * MSKP_SMAX + i].kind == MSKP_KIND_NEXT;    // units that ended at the next restart point
            out6[4] += rs[c * MSKP_SMAX + i].kind == MSKP_KIND_HANDOFF; // ... somewhere else (stale tag, end of the row)
            const long long span = rs[c * MSKP_SMAX + i].end.a - rp[c * MSKP_SMAX + i].relA;
            out6[8] = std::max(out6[8], span); // longest unit, items
            out6[9] += span;
            // links: a unit that ended at the next restart point with exactly the delay registers that one assumed
            if (i + 1 < nrst[c] && rs[c * MSKP_SMAX + i].kind == MSKP_KIND_NEXT) {
                const mskp_res &a = rs[c * MSKP_SMAX + i], &b = rs[c * MSKP_SMAX + i + 1];
                out6[6] += mskp_same_bits(a.end.y, b.ay) && mskp_same_bits(a.end.nl, b.anl);
                out6[7] += 1;
            }
        }
    }
    return AISX_OK;
}

extern "C" int aisx_msk_last_status(aisx_msk* h, int* status, void* stream)
{
    if (!h || !status)
        return AISX_ERR_INVALID;
    std::vector<int> st(h->nchan);
    AISX_HIPCHK(hipMemcpyAsync(st.data(), h->d_status, sizeof(int) * h->nchan, hipMemcpyDeviceToHost,
                               (hipStream_t)stream));
    AISX_HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    int acc = 0;
    for (int v : st)
        acc |= v;
    *status = acc;
    return AISX_OK;
}

// GNU Radio path: the call's control words set, and its three result words gathered behind the symbols' header,
// by one-thread kernels -- a blocking hipMemcpy of four bytes costs as much as a launch, and a call had eight of them
__global__ void k_msk_host_setup(int* tagn, int ntags, unsigned long long* nread, unsigned long long R, int* carry_len, int* ctag_n)
{
    *tagn = ntags;
    *nread = R;
    *carry_len = 0;
    *ctag_n = 0;
}
__global__ void k_msk_host_pack(int* hdr, const int* produced, const int* consumed, const int* status)
{
    hdr[0] = *produced;
    hdr[1] = *consumed;
    hdr[2] = *status;
    hdr[3] = 0;
}

extern "C" int aisx_msk_general_work_host(aisx_msk* h, int noutput_items, int ninput_items, const aisx_cf32* in,
                                          aisx_cf32* out, float* out_err, float* out_mu, uint8_t* out_bits,
                                          const aisx_tag* tags, int ntags, uint64_t nitems_read,
                                          int in_has_lookahead, int* consumed, int* produced)
{
    if (!h || !in || !out || !consumed || !produced || noutput_items < 0 || ninput_items < 0 || ntags < 0)
        return AISX_ERR_INVALID;
    if (h->nchan != 1) {
        set_err("aisx_msk_general_work_host: handle has %d channels, the GNU Radio path needs 1", h->nchan);
        return AISX_ERR_INVALID;
    }
    *consumed = 0;
    *produced = 0;
    if (ninput_items == 0 || noutput_items == 0)
        return AISX_OK;
    int rc;
    // the interpolator reads up to in[ninput_items] (one past): stage one spare item
    const int nin = ninput_items + 1;
    if (nin > h->st_in_cap) {
        h->st_in_cap = 0; // (until the new buffer exists: a failed allocation leaves no stale capacity behind)
        dev_free(h->d_st_in);
        if ((rc = dev_alloc(&h->d_st_in, nin)) != AISX_OK)
            return rc;
        h->st_in_cap = nin;
    }
    if (noutput_items > h->st_out_cap) {
        h->st_out_cap = 0;
        dev_free(h->d_st_blk);
        h->d_st_blk = nullptr;
        h->d_st_sym = nullptr;
        dev_free(h->d_st_err);
        dev_free(h->d_st_mu);
        dev_free(h->d_st_bits);
        // (symbols behind a 16-byte header {produced, consumed, status, 0}: one copy brings back both)
        if ((rc = dev_alloc(&h->d_st_blk, (size_t)noutput_items + 2)) != AISX_OK || (rc = dev_alloc(&h->d_st_err, noutput_items)) != AISX_OK ||
            (rc = dev_alloc(&h->d_st_mu, noutput_items)) != AISX_OK || (rc = dev_alloc(&h->d_st_bits, noutput_items)) != AISX_OK)
            return rc;
        h->d_st_sym = h->d_st_blk + 2;
        h->st_out_cap = noutput_items;
        h->st_host.resize((size_t)noutput_items + 2);
    }
    if (ntags + 1 > h->st_tag_cap) {
        h->st_tag_cap = 0;
        dev_free(h->d_st_tags);
        dev_free(h->d_st_tagn);
        if ((rc = dev_alloc(&h->d_st_tags, ntags + 1)) != AISX_OK || (rc = dev_alloc(&h->d_st_tagn, 1)) != AISX_OK)
            return rc;
        h->st_tag_cap = ntags + 1;
    }
    AISX_HIPCHK(hipMemcpy(h->d_st_in, in, sizeof(cf) * (in_has_lookahead ? nin : ninput_items), hipMemcpyHostToDevice));
    if (!in_has_lookahead)
        AISX_HIPCHK(hipMemset(h->d_st_in + ninput_items, 0, sizeof(cf)));
    if (ntags > 0)
        AISX_HIPCHK(hipMemcpy(h->d_st_tags, tags, sizeof(tag_rec) * ntags, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_msk_host_setup, dim3(1), dim3(1), 0, 0, h->d_st_tagn, ntags, h->d_nread, (unsigned long long)nitems_read,
                       h->d_carry_len[h->cur], h->d_ctag_n[h->cur]);
    AISX_HIPCHK(hipGetLastError());
    if ((rc = msk_launch_tagprep(h, h->d_st_tags, h->d_st_tagn, ntags + 1, 0)) != AISX_OK)
        return rc;
    MskParams p;
    msk_fill_common(h, p);
    p.in = h->d_st_in;
    p.in_stride = nin;
    p.n = ninput_items;
    p.stream_mode = 0;
    p.gr_ninput = ninput_items;
    p.gr_noutput = noutput_items;
    p.syms = h->d_st_sym;
    p.err = h->d_st_err;
    p.mu_out = h->d_st_mu;
    p.out_stride = noutput_items;
    p.sym_al16 = ((uintptr_t)h->d_st_sym % 16 == 0) && (noutput_items % 2 == 0);
    p.out_cap = noutput_items;
    p.produced = h->d_produced;
    if ((rc = msk_launch(p, 1, 0)) != AISX_OK)
        return rc;
    h->cur ^= 1;
    // (the NRZI bit tail only for a caller that takes its output -- the gr::ais block does not, python/ais_demod.py:48-52
    // are blocks of their own there: its state then carries on from the last call that did)
    if (out_bits && (rc = msk_launch_bittail(h, h->d_st_sym, noutput_items, h->d_produced, h->d_st_bits, noutput_items,
                                             noutput_items, 0)) != AISX_OK)
        return rc;
    hipLaunchKernelGGL(k_msk_host_pack, dim3(1), dim3(1), 0, 0, (int*)h->d_st_blk, h->d_produced, h->d_consumed, h->d_status);
    AISX_HIPCHK(hipGetLastError());
    // header + every symbol the call may have produced in one copy (at most noutput_items of them: a few KB)
    AISX_HIPCHK(hipMemcpy(h->st_host.data(), h->d_st_blk, sizeof(cf) * ((size_t)noutput_items + 2), hipMemcpyDeviceToHost));
    const int* hdr = (const int*)h->st_host.data();
    *produced = hdr[0];
    *consumed = hdr[1];
    const int st = hdr[2];
    const int np = *produced;
    if (np > 0) {
        memcpy(out, h->st_host.data() + 2, sizeof(cf) * (size_t)np);
        if (out_err)
            AISX_HIPCHK(hipMemcpy(out_err, h->d_st_err, sizeof(float) * np, hipMemcpyDeviceToHost));
        if (out_mu)
            AISX_HIPCHK(hipMemcpy(out_mu, h->d_st_mu, sizeof(float) * np, hipMemcpyDeviceToHost));
        if (out_bits)
            AISX_HIPCHK(hipMemcpy(out_bits, h->d_st_bits, np, hipMemcpyDeviceToHost));
    }
    if (st & MSK_ST_INTERP_RANGE) {
        set_err("mmse_fir_interpolator_cc: imu out of bounds."); // upstream std::runtime_error
        return AISX_ERR_RUNTIME;
    }
    return AISX_OK;
}
