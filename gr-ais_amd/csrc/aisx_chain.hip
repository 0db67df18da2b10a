// aisx_chain.hip -- C ABI (include/aisx.h) of the pipelined demod chain: the connect order of
// python/ais_demod.py:56 (freq_sync -> agc -> corr_est -> msk timing recovery -> NRZI bit tail)
// driven as one call per step over the stage handles, with the stream / event / buffer
// choreography that makes the stages of neighbouring steps overlap.
// Host code only: every kernel is launched through the stages' own entry points.
#include <stdlib.h>
#include <string>

#include "aisx_host.h"

using namespace aisx;

struct aisx_chain {
    aisx_freqsync* fs = nullptr; // borrowed; null = core chain (corr_est -> msk only)
    aisx_agc* agc = nullptr;     // borrowed
    int agc_claim_prev = -1;     // the handle's LDS claim before this chain set its own (-1: the chain set none)
    int walk_claim_prev = -1;    // ... and the phase walk's (aisx_freqsync_set_walk_lds_claim)
    aisx_corr* corr = nullptr;   // borrowed
    aisx_msk* msk = nullptr;     // borrowed
    int nchan = 0, max_items = 0, fftlen = 0;
    int serial = 0; // AISX_CHAIN_SERIAL: every stage on s_main (A/B runs)
    // streams: sample passes | timing recovery | its bit tail | NCO phase walk one step ahead
    hipStream_t s_main = nullptr, s_msk = nullptr, s_tail = nullptr, s_walk = nullptr;
    long long corr_calls = 0;          // aisx_corr_process calls made so far
    long long corr_call_of[8] = { 0 }; // [step % NBUF]: corr_calls behind that step's call, 0 for a step without one
    bool failed = false;               // a step failed half way: see aisx_chain_step
    int msk_cus = 0; // compute units set aside for the timing recovery's stream (0: streams share the chip)
    static constexpr int NBUF = AISX_CHAIN_DEPTH;
    cf* d_y = nullptr; // front-end output (stock chain): one buffer, written and read on s_main
    long y_stride = 0;
    cf* d_yc[NBUF] = {}; // corr_est's delayed output, read by the recovery of the same step on s_msk
    long yc_stride = 0;
    hipEvent_t ev_in = nullptr;
    hipEvent_t ev_ready[NBUF] = {};    // s_main: the step's sample passes and tags are done (its input is free)
    hipEvent_t ev_msk_done[NBUF] = {}; // s_msk: the step's recovery has read d_yc[par] and its tags
    hipEvent_t ev_done[NBUF] = {};     // the step's outputs are complete (bit tail included)
    long long nsteps = 0;
    int m_of[NBUF] = {}; // items the correlator wrote per row in the step that owns d_yc[k]
    int npend = 0; // items the front end holds back (n % fftlen arithmetic of stream_to_vector)
    // what aisx_freqsync_estimate_ahead was last asked to prepare and has not been consumed yet
    const void* ahead_in = nullptr;
    long ahead_stride = 0;
    int ahead_n = 0;
};

static void chain_free(aisx_chain* h)
{
    if (!h)
        return;
    if (h->msk)
        (void)aisx_msk_set_tail_stream(h->msk, nullptr, 0);
    if (h->agc && h->agc_claim_prev >= 0) // (the placement claims are the chain's: the handles go back as they came)
        (void)aisx_agc_set_lds_claim(h->agc, h->agc_claim_prev);
    if (h->fs && h->walk_claim_prev >= 0)
        (void)aisx_freqsync_set_walk_lds_claim(h->fs, h->walk_claim_prev);
    for (hipStream_t s : { h->s_main, h->s_msk, h->s_tail, h->s_walk })
        if (s)
            (void)hipStreamSynchronize(s);
    dev_free(h->d_y);
    for (int k = 0; k < aisx_chain::NBUF; k++) {
        dev_free(h->d_yc[k]);
        for (hipEvent_t e : { h->ev_ready[k], h->ev_msk_done[k], h->ev_done[k] })
            if (e)
                (void)hipEventDestroy(e);
    }
    if (h->ev_in)
        (void)hipEventDestroy(h->ev_in);
    for (hipStream_t s : { h->s_main, h->s_msk, h->s_tail, h->s_walk })
        if (s)
            (void)hipStreamDestroy(s);
    delete h;
}

// The LDS a front-end (k_agcw) workgroup claims beyond the `used` bytes it needs: how many of them the dispatcher can put
// on a CU beside a timing-recovery workgroup (msk_lds bytes each, msk_wgs of them, one per CU at most), from the part's
// own figures (ncu CUs of lds_cu bytes).  The rule and the sweeps it follows: DESIGN.md 4.8, DESIGN_APPENDIX.md A.6,
// tools/claim_sweep.py, profiles/r06_claim_sweep.json.
//   * the recovery leaves at least half of the CUs free: none beside it (used + claim > what it leaves), two per free CU
//     (63 KB on this part: -4 ... -11 % per step against no claim at 2048 ... 4096 channels);
//   * a recovery workgroup on more than half of the CUs, all of them resident at once: a third of what the recovery
//     leaves (23 KB: two beside each; 6144 / 8192 channels -1 ... -2 %, where 63 KB costs +15 %);
//   * more recovery workgroups than CUs (they come in rounds): two thirds of it (46 KB: one beside each; 12 288 / 16 384
//     channels -3 ... -8 %).
// Unknown figures (a query failed) or a part where the arithmetic does not work out: no claim.
static int chain_front_claim(int ncu, int lds_cu, int msk_wgs, int msk_lds, int used)
{
    if (ncu <= 0 || lds_cu <= 0 || msk_wgs <= 0 || msk_lds <= 0 || used <= 0 || msk_lds >= lds_cu)
        return 0;
    const int left = lds_cu - msk_lds; // beside a recovery workgroup
    const int kb = 1024;
    if (2 * msk_wgs <= ncu) {
        int claim = (left - used) / kb * kb + kb; // the first whole KB with used + claim > left
        if (claim < 0)
            claim = 0;
        return 2 * (used + claim) <= lds_cu ? claim : 0;
    }
    int claim = ((msk_wgs <= ncu ? 1 : 2) * left / 3) / kb * kb;
    if (used + claim > left)
        claim = (left - used) / kb * kb;
    return claim > 0 ? claim : 0;
}

extern "C" int aisx_chain_create(aisx_chain** out, aisx_freqsync* fs, aisx_agc* agc, aisx_corr* corr, aisx_msk* msk,
                                 int nchan, int max_items, int fftlen)
{
    if (!out)
        return AISX_ERR_INVALID;
    *out = nullptr;
    if (!corr || !msk || (fs == nullptr) != (agc == nullptr) || nchan < 1 || max_items < 1 || (fs && fftlen < 1)) {
        set_err("aisx_chain_create: needs corr_est and msk handles, freq_sync and agc both or neither");
        return AISX_ERR_INVALID;
    }
    int rc = require_device();
    if (rc != AISX_OK)
        return rc;
    {
        // the stage handles are borrowed: their channel counts and capacities must cover what the chain
        // will hand them (rows of d_y / d_yc are sized here, the kernels' grids there)
        int nc = 0, mi = 0, fl = 0, W = 0, fused_ok = 0;
        const int need = max_items + (fs ? fftlen : 0); // a step emits every whole vector of (pending + new) items
        bool ok = true;
        char why[160] = "";
        auto bad = [&](const char* stage, const char* what, int got, int want) {
            if (ok)
                snprintf(why, sizeof why, "%s handle: %s %d, the chain needs %d", stage, what, got, want);
            ok = false;
        };
        aisx_corr_geometry(corr, &nc, &mi);
        if (nc != nchan)
            bad("corr_est", "nchan", nc, nchan);
        else if (mi < need)
            bad("corr_est", "max_items", mi, need);
        aisx_msk_geometry(msk, &nc, &mi);
        if (nc != nchan)
            bad("msk_timing_recovery", "nchan", nc, nchan);
        else if (mi < need)
            bad("msk_timing_recovery", "max_items", mi, need);
        if (fs) {
            aisx_freqsync_geometry(fs, &nc, &mi, &fl);
            if (aisx_freqsync_is_estimator_only(fs))
                bad("freq_sync", "made by aisx_freqest_create for the estimator alone, fftlen", fl, 1024);
            else if (nc != nchan)
                bad("freq_sync", "nchan", nc, nchan);
            else if (fl != fftlen)
                bad("freq_sync", "fftlen", fl, fftlen);
            else if (mi < max_items)
                bad("freq_sync", "max_items", mi, max_items);
            aisx_agc_geometry(agc, &nc, &mi, &W, &fused_ok);
            if (nc != nchan)
                bad("agc", "nchan", nc, nchan);
            else if (mi < need)
                bad("agc", "max_items", mi, need);
            else if (!fused_ok)
                bad("agc", "window (a multiple of 8 in [16, 2048] for the fused front end)", W, 512);
        }
        if (!ok) {
            set_err("aisx_chain_create: %s", why);
            return AISX_ERR_INVALID;
        }
    }
    aisx_chain* h = new aisx_chain();
    h->fs = fs;
    h->agc = agc;
    h->corr = corr;
    h->msk = msk;
    h->nchan = nchan;
    h->max_items = max_items;
    h->fftlen = fs ? fftlen : 0;
    if (const char* e = exp_env("AISX_CHAIN_SERIAL"))
        h->serial = atoi(e) != 0;
#define CKH(e)                                                                                     \
    do {                                                                                           \
        hipError_t e__ = (e);                                                                      \
        if (e__ != hipSuccess) {                                                                   \
            set_err("aisx_chain_create: %s failed: %s", #e, hipGetErrorString(e__));               \
            chain_free(h);                                                                         \
            return AISX_ERR_HIP;                                                                   \
        }                                                                                          \
    } while (0)
    {
        // AISX_CHAIN_MSK_CUS = N: the timing recovery's stream owns N compute units (CU-mask bits [0, N): the
        // driver deals mask bits round the XCDs, so N / 8 CUs in each), the other streams the remaining ones
        int ncu = 0, msk_cus = 0, walk_with_msk = 0, tail_with_msk = 0;
        hipDeviceProp_t prop;
        int dev = 0;
        CKH(hipGetDevice(&dev));
        CKH(hipGetDeviceProperties(&prop, dev));
        ncu = prop.multiProcessorCount;
        if (const char* e = exp_env("AISX_CHAIN_MSK_CUS"))
            msk_cus = atoi(e);
        if (const char* e = exp_env("AISX_CHAIN_WALK_WITH_MSK"))
            walk_with_msk = atoi(e);
        if (const char* e = exp_env("AISX_CHAIN_TAIL_WITH_MSK"))
            tail_with_msk = atoi(e);
        if (h->serial || msk_cus < 0 || msk_cus >= ncu)
            msk_cus = 0;
        auto make = [&](hipStream_t* s, int lo, int hi) -> hipError_t {
            if (msk_cus == 0)
                return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
            std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
            for (int b = lo; b < hi; b++)
                mask[(size_t)b / 32] |= 1u << (b % 32);
            return hipExtStreamCreateWithCUMask(s, (uint32_t)mask.size(), mask.data());
        };
        CKH(make(&h->s_main, msk_cus, ncu));
        CKH(make(&h->s_msk, 0, msk_cus));
        CKH(tail_with_msk ? make(&h->s_tail, 0, msk_cus) : make(&h->s_tail, msk_cus, ncu));
        CKH(walk_with_msk ? make(&h->s_walk, 0, msk_cus) : make(&h->s_walk, msk_cus, ncu));
        h->msk_cus = msk_cus;
    }
    CKH(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
    for (int k = 0; k < aisx_chain::NBUF; k++) {
        CKH(hipEventCreateWithFlags(&h->ev_ready[k], hipEventDisableTiming));
        CKH(hipEventCreateWithFlags(&h->ev_msk_done[k], hipEventDisableTiming));
        CKH(hipEventCreateWithFlags(&h->ev_done[k], hipEventDisableTiming));
    }
#undef CKH
    // a step's front end emits every complete fftlen-vector of (pending + new) items
    const long cap = (long)max_items + h->fftlen;
    h->y_stride = h->yc_stride = (cap + 1) & ~1L; // rows 16-byte aligned
    if (fs && (rc = dev_alloc(&h->d_y, (size_t)nchan * h->y_stride, false)) != AISX_OK) {
        chain_free(h);
        return rc;
    }
    for (int k = 0; k < aisx_chain::NBUF; k++)
        if ((rc = dev_alloc(&h->d_yc[k], (size_t)nchan * h->yc_stride, false)) != AISX_OK) {
            chain_free(h);
            return rc;
        }
    // What the part offers and what the recovery kernel's launch takes of it: the two placement decisions below follow.
    int ncu = 0, lds_cu = 0, msk_wgs = 0, msk_lds = 0;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            ncu = prop.multiProcessorCount;
            lds_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
        }
        (void)aisx_msk_placement(msk, &msk_wgs, &msk_lds);
    }
    if (!h->serial && agc) {
        // The front-end kernel's workgroups are placed by an LDS claim (aisx_agc_set_lds_claim, chain_front_claim above):
        // while the recovery leaves half of the chip free (up to 4096 channels on 256 CUs) they stay off the CUs that
        // hold a recovery workgroup and two of them fit a free CU; with a recovery workgroup on (nearly) every CU ONE
        // sits beside each (the sweep: DESIGN_APPENDIX.md A.6 -- a claim too large there makes the front end wait for
        // the recovery to finish: 10.6 against 9.2 ms per step at 8192 channels).
        int prev = 0, used = 0;
        (void)aisx_agc_get_lds_claim(agc, &prev, &used);
        int claim = chain_front_claim(ncu, lds_cu, msk_wgs, msk_lds, used);
        if (const char* e = exp_env("AISX_CHAIN_AGC_CLAIM")) {
            const int v = atoi(e);
            claim = v < 0 ? 0 : (v > 144 * 1024 ? 144 * 1024 : v);
        }
        h->agc_claim_prev = prev;
        if ((rc = aisx_agc_set_lds_claim(agc, claim)) != AISX_OK) {
            chain_free(h);
            return rc;
        }
        // The phase walk's one-wave workgroups (3 KB) likewise, while the recovery leaves half of the CUs free: one of them
        // beside a recovery workgroup delays its recurrence and takes the LDS the correlator's workgroup would have had
        // there (4096 channels: the step 5.43-5.47 against 5.47-5.53 ms, the correlator 1.45-1.47 against 1.52-1.54).
        // With a recovery workgroup on more than half of the CUs the walk has to run beside them: no claim.
        if (fs && ncu > 0 && 2 * msk_wgs <= ncu) {
            int wprev = 0, wused = 0;
            (void)aisx_freqsync_get_walk_lds_claim(fs, &wprev, &wused);
            const int wclaim = chain_front_claim(ncu, lds_cu, msk_wgs, msk_lds, wused);
            h->walk_claim_prev = wprev;
            if ((rc = aisx_freqsync_set_walk_lds_claim(fs, wclaim)) != AISX_OK) {
                chain_free(h);
                return rc;
            }
        }
    }
    if (!h->serial) {
        // the bit tail of step k beside the recovery of step k + 1; the next step's sample passes
        // behind this step's tag prepass (aisx_msk_wait_prepass: the first call arms the event)
        int us = 20; // AISX_MSK_HEADSTART_US: the recovery kernel's head start at the dispatcher (aisx_msk_set_head_start)
        if (const char* e = exp_env("AISX_MSK_HEADSTART_US"))
            us = atoi(e);
        if ((rc = aisx_msk_set_tail_stream(msk, h->s_tail, 1)) != AISX_OK || (rc = aisx_msk_set_head_start(msk, us < 0 ? 0 : us)) != AISX_OK ||
            (rc = aisx_msk_wait_prepass(msk, h->s_main)) != AISX_OK) {
            chain_free(h);
            return rc;
        }
    }
    *out = h;
    return AISX_OK;
}

extern "C" int aisx_chain_destroy(aisx_chain* h)
{
    chain_free(h);
    return AISX_OK;
}

extern "C" int aisx_chain_depth(void) { return aisx_chain::NBUF; }

static int chain_step_issue(aisx_chain* h, const aisx_cf32* d_in, long in_stride, int n, const aisx_cf32* d_in_next, long next_stride,
                            int n_next, aisx_cf32* d_syms, uint8_t* d_bits, long out_stride, int* d_produced, void* stream,
                            long long* step);

extern "C" int aisx_chain_step(aisx_chain* h, const aisx_cf32* d_in, long in_stride, int n, const aisx_cf32* d_in_next,
                               long next_stride, int n_next, aisx_cf32* d_syms, uint8_t* d_bits, long out_stride,
                               int* d_produced, void* stream, long long* step)
{
    if (!h || !d_in || n < 1 || n > h->max_items || in_stride < n || !d_produced ||
        (d_in_next && (n_next < 1 || n_next > h->max_items || next_stride < n_next))) {
        set_err("aisx_chain_step: bad argument (n = %d, max_items = %d)", n, h ? h->max_items : -1);
        return AISX_ERR_INVALID;
    }
    // (what only the last stage would notice: checked here, before the first stage is issued, so that an
    // argument error leaves the chain and the stages' histories as they were)
    if (out_stride < 1 || out_stride >= (1L << 23)) {
        set_err("aisx_chain_step: out_stride %ld outside 1 .. 2^23 - 1", out_stride);
        return AISX_ERR_INVALID;
    }
    if (h->failed) {
        set_err("aisx_chain_step: an earlier step failed half way (its stages had been issued in part): the stage handles' "
                "streams and histories no longer match; aisx_*_reset the stages and create a new chain");
        return AISX_ERR_INVALID;
    }
    const int rc = chain_step_issue(h, d_in, in_stride, n, d_in_next, next_stride, n_next, d_syms, d_bits, out_stride, d_produced, stream,
                                    step);
    if (rc != AISX_OK) {
        // Not transactional: stages issued before the failure have run (histories advanced, preparations
        // queued).  What was prepared ahead is dropped, and the chain refuses further steps.
        const std::string msg = aisx_last_error();
        if (h->fs)
            aisx_freqsync_drop_ahead(h->fs, h->s_main);
        h->ahead_in = nullptr;
        h->failed = true;
        set_err("%s", msg.c_str());
    }
    return rc;
}

static int chain_step_issue(aisx_chain* h, const aisx_cf32* d_in, long in_stride, int n, const aisx_cf32* d_in_next, long next_stride,
                            int n_next, aisx_cf32* d_syms, uint8_t* d_bits, long out_stride, int* d_produced, void* stream,
                            long long* step)
{
    int rc;
    const int par = (int)(h->nsteps % aisx_chain::NBUF);
    hipStream_t sm = h->s_main, sk = h->serial ? h->s_main : h->s_msk;
    // the caller's stream has produced d_in (and d_in_next)
    AISX_HIPCHK(hipEventRecord(h->ev_in, (hipStream_t)stream));
    AISX_HIPCHK(hipStreamWaitEvent(sm, h->ev_in, 0));
    if (h->nsteps >= aisx_chain::NBUF) // step k - NBUF has released d_yc[par] and its tags
        AISX_HIPCHK(hipStreamWaitEvent(sm, h->ev_msk_done[par], 0));

    const cf* y = (const cf*)d_in;
    long ys = in_stride;
    int m = n;
    if (h->fs) {
        // The frequency estimates and the NCO phase walk of step k + 1 are issued BEFORE the
        // front-end pass of step k when two preparations may wait (whole vectors, nothing pending):
        // the estimates on s_main (kernels with large grids are dispatched one after the other
        // whatever their streams), the walk -- a strict recurrence, one lane per channel -- on
        // s_walk, beside this step's sample passes.  Otherwise behind the pass.
        const bool prepared = h->ahead_in == (const void*)d_in && h->ahead_stride == in_stride && h->ahead_n == n;
        // (a preparation for other arguments is still queued in the freq_sync handle: the pass below
        // drops it and estimates for itself; nothing may be queued behind it before that)
        const bool stale = h->ahead_in != nullptr && !prepared;
        const bool early = !h->serial && !stale && h->npend == 0 && n % h->fftlen == 0;
        hipStream_t sw = h->serial ? sm : h->s_walk;
        if (early) {
            if (!prepared && (rc = aisx_freqsync_estimate_ahead(h->fs, d_in, in_stride, n, sm, sw)) != AISX_OK)
                return rc;
            if (d_in_next && (rc = aisx_freqsync_estimate_ahead(h->fs, d_in_next, next_stride, n_next, sm, sw)) != AISX_OK)
                return rc;
        }
        int nout = 0;
        if ((rc = aisx_freqsync_agc_process(h->fs, h->agc, d_in, in_stride, n, (aisx_cf32*)h->d_y, h->y_stride, nullptr, 0, &nout,
                                            sm)) != AISX_OK)
            return rc;
        h->npend = h->npend + n - nout;
        h->ahead_in = nullptr;
        if (!early && !h->serial && d_in_next &&
            (rc = aisx_freqsync_estimate_ahead(h->fs, d_in_next, next_stride, n_next, sm, sw)) != AISX_OK)
            return rc;
        if (d_in_next && !h->serial) {
            h->ahead_in = d_in_next;
            h->ahead_stride = next_stride;
            h->ahead_n = n_next;
        }
        y = h->d_y;
        ys = h->y_stride;
        m = nout;
    }
    if (m == 0) { // not one whole vector yet: nothing reaches the correlator (stream_to_vector holds the items)
        AISX_HIPCHK(hipMemsetAsync(d_produced, 0, sizeof(int) * h->nchan, sm));
        // (the events of this slot are recorded afresh on s_main: behind what they stood for -- step k - NBUF's
        // bit tail -- so that a wait for that step through the reused slot still holds)
        if (h->nsteps >= aisx_chain::NBUF)
            AISX_HIPCHK(hipStreamWaitEvent(sm, h->ev_done[par], 0));
        h->corr_call_of[par] = 0;
        AISX_HIPCHK(hipEventRecord(h->ev_ready[par], sm));
        AISX_HIPCHK(hipEventRecord(h->ev_msk_done[par], sm));
        AISX_HIPCHK(hipEventRecord(h->ev_done[par], sm));
    } else {
        if ((rc = aisx_corr_process(h->corr, (const aisx_cf32*)y, ys, (aisx_cf32*)h->d_yc[par], h->yc_stride, nullptr, 0, m, sm)) != AISX_OK)
            return rc;
        h->corr_call_of[par] = ++h->corr_calls;
        const aisx_tag* tags = nullptr;
        const int* counts = nullptr;
        int tcap = 0;
        if ((rc = aisx_corr_tags_device(h->corr, &tags, &counts, &tcap)) != AISX_OK)
            return rc;
        AISX_HIPCHK(hipEventRecord(h->ev_ready[par], sm));
        if (sk != sm)
            AISX_HIPCHK(hipStreamWaitEvent(sk, h->ev_ready[par], 0));
        // (ev_ready: what the time-parallel recovery's units wait for on their own stream -- they need this
        // step's samples and tags, not the previous step's recovery, and run beside it)
        if ((rc = aisx_msk_process_stream_after(h->msk, (const aisx_cf32*)h->d_yc[par], h->yc_stride, m, tags, counts, tcap, d_syms,
                                                nullptr, nullptr, d_bits, out_stride, d_produced, sk,
                                                h->serial ? nullptr : (void*)h->ev_ready[par])) != AISX_OK)
            return rc;
        AISX_HIPCHK(hipEventRecord(h->ev_msk_done[par], sk));
        // the bit tail (if any) was queued on s_tail behind the recovery
        AISX_HIPCHK(hipEventRecord(h->ev_done[par], (d_bits && !h->serial) ? h->s_tail : sk));
        // the next step's sample passes start behind this step's tag prepass, i.e. when the recovery
        // kernel stands at the head of its queue (aisx_msk_wait_prepass; include/aisx.h)
        if (!h->serial && (rc = aisx_msk_wait_prepass(h->msk, sm)) != AISX_OK)
            return rc;
    }
    h->m_of[par] = m;
    if (step)
        *step = h->nsteps;
    h->nsteps++;
    return AISX_OK;
}

static int chain_wait_event(aisx_chain* h, long long step, void* stream, bool have_stream, hipEvent_t* evs, const char* what)
{
    if (!h || step < 0 || step >= h->nsteps) {
        set_err("%s: step %lld has not been issued", what, step);
        return AISX_ERR_INVALID;
    }
    // (an event already reused by a later step of the same parity completes behind the step asked
    // for: waiting for it is still correct)
    hipEvent_t e = evs[step % aisx_chain::NBUF];
    if (have_stream)
        AISX_HIPCHK(hipStreamWaitEvent((hipStream_t)stream, e, 0));
    else
        AISX_HIPCHK(hipEventSynchronize(e));
    return AISX_OK;
}

extern "C" int aisx_chain_wait(aisx_chain* h, long long step, void* stream, int host_blocks)
{
    return chain_wait_event(h, step, stream, !host_blocks, h ? h->ev_done : nullptr, "aisx_chain_wait");
}

extern "C" int aisx_chain_wait_input(aisx_chain* h, long long step, void* stream, int host_blocks)
{
    return chain_wait_event(h, step, stream, !host_blocks, h ? h->ev_ready : nullptr, "aisx_chain_wait_input");
}

extern "C" int aisx_chain_read_corr_output(aisx_chain* h, long long step, int chan0, int nch, aisx_cf32* d_dst, long dst_stride,
                                           int* n, void* stream)
{
    if (!h || step < 0 || step >= h->nsteps || h->nsteps - step > aisx_chain::NBUF || chan0 < 0 || nch < 1 ||
        chan0 + nch > h->nchan || !d_dst) {
        set_err("aisx_chain_read_corr_output: step %lld is not among the last %d, or bad channel range", step, aisx_chain::NBUF);
        return AISX_ERR_INVALID;
    }
    const int par = (int)(step % aisx_chain::NBUF), m = h->m_of[par];
    if (n)
        *n = m;
    if (m == 0)
        return AISX_OK;
    if (dst_stride < m)
        return AISX_ERR_INVALID;
    AISX_HIPCHK(hipStreamWaitEvent((hipStream_t)stream, h->ev_ready[par], 0));
    AISX_HIPCHK(hipMemcpy2DAsync(d_dst, sizeof(cf) * dst_stride, h->d_yc[par] + (size_t)chan0 * h->yc_stride, sizeof(cf) * h->yc_stride,
                                 sizeof(cf) * m, nch, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return AISX_OK;
}

extern "C" int aisx_chain_read_tags(aisx_chain* h, long long step, aisx_tag* host_tags, int host_cap, int* ntags, void* stream)
{
    if (!h || !ntags || step < 0 || step >= h->nsteps || h->nsteps - step > aisx_chain::NBUF) {
        set_err("aisx_chain_read_tags: step %lld is not among the last %d issued", step, aisx_chain::NBUF);
        return AISX_ERR_INVALID;
    }
    *ntags = 0;
    const long long call = h->corr_call_of[step % aisx_chain::NBUF];
    if (call == 0) // that step's front end emitted no whole vector: corr_est was not called
        return AISX_OK;
    const long long back = h->corr_calls - call;
    if (back > 2) { // (cannot happen while NBUF <= 3: every later step made at most one call)
        set_err("aisx_chain_read_tags: the tags of step %lld have been overwritten", step);
        return AISX_ERR_INVALID;
    }
    return aisx_corr_read_tags_back(h->corr, (int)back, host_tags, host_cap, ntags, stream);
}

extern "C" int aisx_chain_synchronize(aisx_chain* h)
{
    if (!h)
        return AISX_ERR_INVALID;
    for (hipStream_t s : { h->s_main, h->s_walk, h->s_msk, h->s_tail })
        AISX_HIPCHK(hipStreamSynchronize(s));
    return AISX_OK;
}

extern "C" void* aisx_chain_stream(aisx_chain* h, int which)
{
    if (!h)
        return nullptr;
    switch (which) {
    case 0: return (void*)h->s_main;
    case 1: return (void*)h->s_msk;
    case 2: return (void*)h->s_tail;
    case 3: return (void*)h->s_walk;
    default: return nullptr;
    }
}
