// aisx_stages.hip -- C ABI for the stages either side of the correlator in the
// reference's chain (python/ais_demod.py:56): square_and_fft_sync_cc / freqest
// (python/gmsk_sync.py, lib/freqest_impl.cc) and analog.feedforward_agc_cc.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "aisx_devctx.h"
#include "aisx_host.h"
#include "aisx_tables.h"
#include "k_agc.h"
#include "k_agcw.h"
#include "k_freqsync.h"

using namespace aisx;

__global__ __launch_bounds__(FS_T) void k_fs_est(FsEstParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
    fs_est_body(cx, p);
}
__global__ __launch_bounds__(FSM_T) void k_fs_mix(FsMixParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
    fs_mix_body(cx, p);
}
__global__ __launch_bounds__(FSW_T) void k_fs_walk(FsWalkParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
    // a strict recurrence on one wave per 64 channels, meant to run beside full-occupancy kernels
    // of other streams: its few instructions go first on their SIMD (as the timing recovery's do)
#ifndef WALK_PRIO
#define WALK_PRIO 3
#endif
    __builtin_amdgcn_s_setprio(WALK_PRIO);
    fs_walk_body(cx, p);
}
__global__ __launch_bounds__(64) void k_fs_freqest(FsFreqestParams p)
{
    DevCtx cx{ nullptr };
    fs_freqest_body(cx, p);
}
// (two workgroups of sixteen waves per CU = eight waves per SIMD: at most 64 VGPRs)
__global__ __launch_bounds__(AGC8_T) void k_agc8(AgcParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
    agc8_body(cx, p);
}

// the streaming form for the stock window (k_agcw.h): four independent waves per workgroup
template <bool MIXED>
__global__ __launch_bounds__(AGW_T) void k_agcw(AgcParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
    agcw_body<MIXED>(cx, p);
}

// exhaustive check of agcw_gain_fast against the float division (k_agcw.h): every float m in
// [AGW_RCP_LO, AGW_RCP_HI], one per lane and grid-stride step
__global__ __launch_bounds__(256) void k_agc_rcp_sweep(float reference, unsigned lo_bits, unsigned hi_bits,
                                                      unsigned long long* count, unsigned* example)
{
    DevCtx cx{ nullptr };
    unsigned long long bad = 0;
    for (unsigned long long b = (unsigned long long)lo_bits + (unsigned long long)blockIdx.x * 256 + threadIdx.x; b <= hi_bits;
         b += (unsigned long long)gridDim.x * 256) {
        const float m = __uint_as_float((unsigned)b);
        const float fast = agcw_gain_fast(cx, reference, m), exact = fdiv_rn(reference, m);
        if (__float_as_uint(fast) != __float_as_uint(exact)) {
            bad++;
            atomicMax(example, (unsigned)b);
        }
    }
    if (bad)
        atomicAdd(count, bad);
}

extern "C" int aisx_util_agc_rcp_mismatches(float reference, unsigned long long* count, float* example)
{
    if (!count)
        return AISX_ERR_INVALID;
    int rc = require_device();
    if (rc != AISX_OK)
        return rc;
    if (!agcw_fast_reference(reference)) {
        set_err("aisx_util_agc_rcp_mismatches: %g is not a reference the reciprocal form serves", reference);
        return AISX_ERR_INVALID;
    }
    unsigned long long* d_count = nullptr;
    unsigned* d_ex = nullptr;
    if ((rc = dev_alloc(&d_count, 1)) != AISX_OK || (rc = dev_alloc(&d_ex, 1)) != AISX_OK) {
        dev_free(d_count);
        return rc;
    }
    unsigned lo, hi, ex = 0;
    const float flo = AGW_RCP_LO, fhi = AGW_RCP_HI;
    memcpy(&lo, &flo, 4);
    memcpy(&hi, &fhi, 4);
    hipLaunchKernelGGL(k_agc_rcp_sweep, dim3(256 * 32), dim3(256), 0, 0, reference, lo, hi, d_count, d_ex);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpy(count, d_count, sizeof(*count), hipMemcpyDeviceToHost);
    if (e == hipSuccess)
        e = hipMemcpy(&ex, d_ex, sizeof(ex), hipMemcpyDeviceToHost);
    dev_free(d_count);
    dev_free(d_ex);
    if (e != hipSuccess) {
        set_err("aisx_util_agc_rcp_mismatches: %s", hipGetErrorString(e));
        return AISX_ERR_HIP;
    }
    if (example)
        memcpy(example, &ex, 4);
    return AISX_OK;
}

__global__ __launch_bounds__(AGC_T) void k_agc(AgcParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
    agc_body(cx, p);
}

// ---------------------------------------------------------------------------
// GNU Radio path staging: device buffers the *_work_host calls copy through, kept between calls and only ever
// grown (a scheduler calls work() thousands of times a second with similar sizes)
template <class T>
static int stage_grow(T** buf, size_t* cap, size_t need)
{
    if (need <= *cap)
        return AISX_OK;
    dev_free(*buf);
    *buf = nullptr;
    *cap = 0;
    const int rc = dev_alloc(buf, need, false);
    if (rc == AISX_OK)
        *cap = need;
    return rc;
}

struct aisx_freqsync {
    cf *d_hst_in = nullptr, *d_hst_out = nullptr; // aisx_freqsync_work_host
    float* d_hst_fh = nullptr;
    size_t hst_in_cap = 0, hst_out_cap = 0, hst_fh_cap = 0;
    int nchan = 0, fftlen = 0, max_items = 0, offset = 0, max_vec = 0;
    // made by aisx_freqest_create with a vector length the freq_sync kernels do not implement: the handle serves
    // aisx_freqest_work / aisx_freqest_work_host only (the search over bins needs no transform of ours)
    bool est_only = false;
    // aisx_freqsync_set_walk_lds_claim: LDS a workgroup of the phase walk claims beyond the 3 KB it uses, and what this
    // handle has raised the kernel's dynamic-LDS limit to
    int walk_claim = 0, walk_attr = 64 * 1024;
    float binsize = 0, sensitivity = 0;
    cf* d_pend[2] = { nullptr, nullptr };
    int cur = 0, npend = 0;
    cf* d_wtab = nullptr;
    int* d_maxpos = nullptr;   // = slot[0].d_maxpos (the two-pass path)
    float* d_phase = nullptr;  // the committed NCO phase, = d_phase3[phase_cur]
    // (three copies in rotation: the committed one and the end phases of up to two walks prepared ahead)
    float* d_phase3[3] = { nullptr, nullptr, nullptr };
    int phase_cur = 0;
    // Fused front end (aisx_freqsync_agc_process).  What a call's sample pass needs from the
    // frequency estimator -- maxpos per vector, the walked NCO phases phi[c][i] (4 bytes per sample,
    // allocated on first use), f-hat -- lives in one of two slots, so that the estimate for call
    // k + 1 can be prepared on another stream (aisx_freqsync_estimate_ahead) while call k's pass
    // still reads its own.
    struct Slot {
        int* d_maxpos = nullptr;
        float* d_phases = nullptr; // phi[c][8 j]: every eighth phase of the walk (FSW_CK)
        float* d_dvec = nullptr;   // the phase increment of each vector
        float* d_fhat = nullptr;
        hipEvent_t ev_read = nullptr; // the last sample pass that read this slot
        bool read_pending = false;
        hipEvent_t ev_ready = nullptr; // behind the walk that filled this slot (estimates prepared ahead)
    } slot[2];
    long phases_stride = 0;
    int slot_cur = 0; // the slot the next process call uses
    hipEvent_t ev_walk = nullptr, ev_proc = nullptr, ev_est = nullptr;
    bool walk_pending = false, proc_pending = false;
    // estimates prepared ahead, in call order: [0] for the next aisx_freqsync_agc_process call (in
    // slot[slot_cur]), [1] for the one after (slot[slot_cur ^ 1]), each for exactly these arguments
    struct Ahead {
        const void* in = nullptr;
        long stride = 0;
        int n = 0;
    } ahead_q[2];
    int ahead_cnt = 0;
    float* d_sintab = nullptr; // gr::fxpt's sine table
    // GNU Radio path staging (aisx_freqest_work_host)
    cf* d_st_vec = nullptr;
    float* d_st_out = nullptr;
    int st_cap = 0;
};

static int fs_whole(const aisx_freqsync* h, const char* who)
{
    if (h && h->est_only) {
        set_err("%s: this handle was made by aisx_freqest_create with fftlen %d and serves aisx_freqest_work / "
                "aisx_freqest_work_host only (square_and_fft_sync_cc is implemented for fftlen = %d)", who, h->fftlen, FS_F);
        return AISX_ERR_INVALID;
    }
    return AISX_OK;
}

extern "C" int aisx_freqsync_create(aisx_freqsync** out, double samplerate, double bits_per_sec, int fftlen, int nchan,
                                    int max_items)
{
    if (!out)
        return AISX_ERR_INVALID;
    *out = nullptr;
    if (nchan < 1 || max_items < 1 || !(samplerate > 0) || !(bits_per_sec > 0)) {
        set_err("aisx_freqsync_create: bad argument");
        return AISX_ERR_INVALID;
    }
    if (fftlen != FS_F) {
        set_err("aisx_freqsync_create: fftlen %d not supported (the gfx950 kernel implements fftlen = %d, the value "
                "python/radio.py:60 uses)", fftlen, FS_F);
        return AISX_ERR_INVALID;
    }
    int rc = require_device();
    if (rc != AISX_OK)
        return rc;
    aisx_freqsync* h = new aisx_freqsync();
    h->nchan = nchan;
    h->fftlen = fftlen;
    h->max_items = max_items;
    // ais.freqest(int(samplerate), int(bits_per_sec), fftlen)  (gmsk_sync.py:25; freqest_impl.cc:46-47)
    const float sr = (float)(int)samplerate;
    const int dr = (int)bits_per_sec;
    h->offset = (int)(fftlen * ((float)dr / sr));
    h->binsize = sr / (float)fftlen;
    // frequency_modulator_fc(-1.0/(float(samplerate)/(2*pi)))  (gmsk_sync.py:27)
    h->sensitivity = (float)(-1.0 / (samplerate / (2 * M_PI)));
    h->max_vec = (max_items + fftlen) / fftlen + 1;
    std::vector<cf> w(FS_F);
    for (int k = 0; k < FS_F; k++) {
        double a = -2.0 * M_PI * (double)k / (double)FS_F;
        w[k] = mk((float)cos(a), (float)sin(a));
    }
#define CK(e)               \
    do {                    \
        rc = (e);           \
        if (rc != AISX_OK) { \
            aisx_freqsync_destroy(h); \
            return rc;      \
        }                   \
    } while (0)
    CK(dev_alloc(&h->d_pend[0], (size_t)nchan * fftlen));
    CK(dev_alloc(&h->d_pend[1], (size_t)nchan * fftlen));
    CK(dev_alloc(&h->d_wtab, FS_F));
    CK(dev_alloc(&h->slot[0].d_maxpos, (size_t)nchan * h->max_vec));
    h->d_maxpos = h->slot[0].d_maxpos;
    for (int k = 0; k < 3; k++)
        CK(dev_alloc(&h->d_phase3[k], nchan));
    h->d_phase = h->d_phase3[0];
#undef CK
    if ((rc = dev_alloc(&h->d_sintab, NCO_TAB_FLOATS)) != AISX_OK) {
        aisx_freqsync_destroy(h);
        return rc;
    }
    if (hipMemcpy(h->d_wtab, w.data(), sizeof(cf) * FS_F, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->d_sintab, aisx_sine_table, sizeof(float) * NCO_TAB_FLOATS, hipMemcpyHostToDevice) != hipSuccess) {
        set_err("aisx_freqsync_create: table upload failed");
        aisx_freqsync_destroy(h);
        return AISX_ERR_HIP;
    }
    *out = h;
    return AISX_OK;
}

extern "C" int aisx_freqsync_geometry(const aisx_freqsync* h, int* nchan, int* max_items, int* fftlen)
{
    if (!h)
        return AISX_ERR_INVALID;
    if (nchan)
        *nchan = h->nchan;
    if (max_items)
        *max_items = h->max_items;
    if (fftlen)
        *fftlen = h->fftlen;
    return AISX_OK;
}

// forget what aisx_freqsync_estimate_ahead has queued (nothing of it was committed); `stream` = where the
// next pass will run: it waits for the walks the dropped preparations still have in flight
extern "C" int aisx_freqsync_drop_ahead(aisx_freqsync* h, void* stream)
{
    if (!h)
        return AISX_ERR_INVALID;
    if (h->ahead_cnt > 0) {
        h->ahead_cnt = 0;
        if (h->walk_pending)
            AISX_HIPCHK(hipStreamWaitEvent((hipStream_t)stream, h->ev_walk, 0));
    }
    return AISX_OK;
}

extern "C" int aisx_freqest_create_n(aisx_freqsync** out, float sample_rate, int data_rate, int fftlen, int nchan, int max_vectors)
{
    // freqest::make(float sample_rate, int data_rate, int fftlen) (include/ais/freqest.h:46): the block alone,
    // with a sample rate that need not be a whole number (lib/freqest_impl.cc:46-47 keep the float) and ANY vector
    // length: work() is a search over fftlen - offset bins of spectra somebody else transformed (:57-88)
    if (out)
        *out = nullptr;
    if (!out || max_vectors < 1 || data_rate < 1 || fftlen < 2 || nchan < 1 || !(sample_rate > 0)) {
        set_err("aisx_freqest_create: bad argument (fftlen %d, data_rate %d, %d channels)", fftlen, data_rate, nchan);
        return AISX_ERR_INVALID;
    }
    int rc;
    if (fftlen == FS_F) {
        if ((rc = aisx_freqsync_create(out, (double)sample_rate, (double)data_rate, fftlen, nchan, max_vectors * fftlen)) != AISX_OK)
            return rc;
    } else {
        if ((rc = require_device()) != AISX_OK)
            return rc;
        aisx_freqsync* h = new aisx_freqsync();
        h->est_only = true;
        h->nchan = nchan;
        h->fftlen = fftlen;
        h->max_items = max_vectors * fftlen;
        h->max_vec = max_vectors;
        *out = h;
    }
    (*out)->offset = (int)(fftlen * ((float)data_rate / sample_rate));
    (*out)->binsize = sample_rate / (float)fftlen;
    return AISX_OK;
}

extern "C" int aisx_freqest_create(aisx_freqsync** out, float sample_rate, int data_rate, int fftlen, int max_vectors)
{
    return aisx_freqest_create_n(out, sample_rate, data_rate, fftlen, 1, max_vectors);
}

extern "C" int aisx_freqsync_is_estimator_only(const aisx_freqsync* h) { return h && h->est_only ? 1 : 0; }

extern "C" int aisx_freqsync_destroy(aisx_freqsync* h)
{
    if (!h)
        return AISX_OK;
    dev_free(h->d_pend[0]);
    dev_free(h->d_pend[1]);
    dev_free(h->d_wtab);
    for (int k = 0; k < 2; k++) {
        dev_free(h->slot[k].d_maxpos);
        dev_free(h->slot[k].d_phases);
        dev_free(h->slot[k].d_dvec);
        dev_free(h->slot[k].d_fhat);
        if (h->slot[k].ev_read)
            (void)hipEventDestroy(h->slot[k].ev_read);
        if (h->slot[k].ev_ready)
            (void)hipEventDestroy(h->slot[k].ev_ready);
    }
    for (int k = 0; k < 3; k++)
        dev_free(h->d_phase3[k]);
    for (hipEvent_t e : { h->ev_walk, h->ev_proc, h->ev_est })
        if (e)
            (void)hipEventDestroy(e);
    dev_free(h->d_st_vec);
    dev_free(h->d_st_out);
    dev_free(h->d_hst_in);
    dev_free(h->d_hst_out);
    dev_free(h->d_hst_fh);
    dev_free(h->d_sintab);
    delete h;
    return AISX_OK;
}

extern "C" int aisx_freqsync_reset(aisx_freqsync* h)
{
    if (!h)
        return AISX_ERR_INVALID;
    if (h->est_only)
        return AISX_OK; // (freqest::work carries nothing from call to call)
    AISX_HIPCHK(hipDeviceSynchronize());
    AISX_HIPCHK(hipMemset(h->d_phase, 0, sizeof(float) * h->nchan));
    AISX_HIPCHK(hipDeviceSynchronize()); // (null-stream fill vs. the caller's non-blocking streams)
    h->npend = 0;
    h->cur = 0;
    h->ahead_cnt = 0;
    h->walk_pending = h->proc_pending = false;
    h->slot[0].read_pending = h->slot[1].read_pending = false;
    return AISX_OK;
}

extern "C" int aisx_freqsync_process(aisx_freqsync* h, const aisx_cf32* d_in, long in_stride, int n, aisx_cf32* d_out,
                                     long out_stride, float* d_fhat, long fhat_stride, int* n_out, void* stream)
{
    if (fs_whole(h, "aisx_freqsync_process") != AISX_OK)
        return AISX_ERR_INVALID;
    if (!h || !d_in || !d_out || !n_out || n < 1 || n > h->max_items || in_stride < n) {
        set_err("aisx_freqsync_process: bad argument");
        return AISX_ERR_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nvec = (h->npend + n) / h->fftlen;
    if (out_stride < (long)nvec * h->fftlen || (d_fhat && fhat_stride < nvec)) {
        set_err("aisx_freqsync_process: output stride too small for %d vectors", nvec);
        return AISX_ERR_INVALID;
    }
    h->ahead_cnt = 0; // (estimates prepared for the fused call are dropped: nothing of them was committed)
    if (h->walk_pending)
        AISX_HIPCHK(hipStreamWaitEvent(st, h->ev_walk, 0));
    if (h->proc_pending)
        AISX_HIPCHK(hipStreamWaitEvent(st, h->ev_proc, 0));
    if (nvec > 0) {
        FsEstParams e;
        e.in = (const cf*)d_in;
        e.in_stride = in_stride;
        e.pend = h->d_pend[h->cur];
        e.npend = h->npend;
        e.wtab = h->d_wtab;
        e.maxpos = h->d_maxpos;
        e.maxpos_stride = h->max_vec;
        e.nvec = nvec;
        e.offset = h->offset;
        hipLaunchKernelGGL(k_fs_est, dim3((nvec + FS_VEC_PER_WG - 1) / FS_VEC_PER_WG, h->nchan), dim3(FS_T), FS_LDS_BYTES, st, e);
        AISX_HIPCHK(hipGetLastError());
    }
    FsMixParams m;
    m.nchan = h->nchan;
    m.in = (const cf*)d_in;
    m.in_stride = in_stride;
    m.pend_in = h->d_pend[h->cur];
    m.pend_out = h->d_pend[h->cur ^ 1];
    m.npend = h->npend;
    m.n = n;
    m.out = (cf*)d_out;
    m.out_stride = out_stride;
    m.maxpos = h->d_maxpos;
    m.maxpos_stride = h->max_vec;
    m.fhat = d_fhat;
    m.fhat_stride = fhat_stride;
    m.phase = h->d_phase;
    m.nvec = nvec;
    m.binsize = h->binsize;
    m.sensitivity = h->sensitivity;
    m.sintab = h->d_sintab;
    hipLaunchKernelGGL(k_fs_mix, dim3((h->nchan + FSM_CPW - 1) / FSM_CPW), dim3(FSM_T), FSM_LDS_BYTES, st, m);
    AISX_HIPCHK(hipGetLastError());
    if (h->ev_proc) {
        // a handle that has been driven through the fused entry points on other streams: what they
        // wait for before touching the pending items, slot 0's maxpos and the phase is THIS pass now
        AISX_HIPCHK(hipEventRecord(h->ev_proc, st));
        AISX_HIPCHK(hipEventRecord(h->ev_walk, st));
        h->proc_pending = h->walk_pending = true;
        AISX_HIPCHK(hipEventRecord(h->slot[0].ev_read, st));
        h->slot[0].read_pending = true;
    }
    h->npend = h->npend + n - nvec * h->fftlen;
    h->cur ^= 1;
    *n_out = nvec * h->fftlen;
    return AISX_OK;
}

extern "C" int aisx_freqest_work(aisx_freqsync* h, const aisx_cf32* d_vecs, long vec_stride, float* d_out,
                                 long out_stride, int nvec, void* stream)
{
    if (!h || !d_vecs || !d_out || nvec < 0 || vec_stride < (long)nvec * h->fftlen || out_stride < nvec)
        return AISX_ERR_INVALID;
    if (nvec == 0)
        return AISX_OK;
    FsFreqestParams p;
    p.vecs = (const cf*)d_vecs;
    p.vec_stride = vec_stride;
    p.out = d_out;
    p.out_stride = out_stride;
    p.nvec = nvec;
    p.fftlen = h->fftlen;
    p.offset = h->offset;
    p.binsize = h->binsize;
    hipLaunchKernelGGL(k_fs_freqest, dim3(h->nchan), dim3(64), 0, (hipStream_t)stream, p);
    AISX_HIPCHK(hipGetLastError());
    return AISX_OK;
}

// square_and_fft_sync_cc as one GNU Radio block (python/gmsk_sync.py:14-37 is a hier block of
// stock blocks around ais.freqest): host pointers, nchan == 1.  `in` = n new items; `out`
// receives every complete fftlen-vector's worth (pending items are kept in the handle, as
// stream_to_vector does); returns the items written.
extern "C" int aisx_freqsync_work_host(aisx_freqsync* h, const aisx_cf32* in, int n, aisx_cf32* out, int out_cap,
                                       float* fhat, int fhat_cap)
{
    if (fs_whole(h, "aisx_freqsync_work_host") != AISX_OK)
        return AISX_ERR_INVALID;
    if (!h || !in || !out || n < 1 || n > h->max_items)
        return AISX_ERR_INVALID;
    if (h->nchan != 1) {
        set_err("aisx_freqsync_work_host: handle has %d channels, the GNU Radio path needs 1", h->nchan);
        return AISX_ERR_INVALID;
    }
    const int nvec = (h->npend + n) / h->fftlen;
    if (out_cap < nvec * h->fftlen || (fhat && fhat_cap < nvec)) {
        set_err("aisx_freqsync_work_host: output buffer too small for %d vectors", nvec);
        return AISX_ERR_INVALID;
    }
    int rc = stage_grow(&h->d_hst_in, &h->hst_in_cap, (size_t)n);
    if (rc == AISX_OK)
        rc = stage_grow(&h->d_hst_out, &h->hst_out_cap, (size_t)nvec * h->fftlen + 1);
    if (rc == AISX_OK && fhat)
        rc = stage_grow(&h->d_hst_fh, &h->hst_fh_cap, (size_t)nvec + 1);
    cf *d_in = h->d_hst_in, *d_out = h->d_hst_out;
    float* d_fh = fhat ? h->d_hst_fh : nullptr;
    int nout = 0;
    if (rc == AISX_OK && hipMemcpy(d_in, in, sizeof(cf) * n, hipMemcpyHostToDevice) != hipSuccess)
        rc = AISX_ERR_HIP;
    if (rc == AISX_OK)
        rc = aisx_freqsync_process(h, (const aisx_cf32*)d_in, n, n, (aisx_cf32*)d_out, (long)nvec * h->fftlen + 1, d_fh, nvec + 1,
                                   &nout, nullptr);
    if (rc == AISX_OK && nout > 0 && hipMemcpy(out, d_out, sizeof(cf) * nout, hipMemcpyDeviceToHost) != hipSuccess)
        rc = AISX_ERR_HIP;
    if (rc == AISX_OK && fhat && nvec > 0 && hipMemcpy(fhat, d_fh, sizeof(float) * nvec, hipMemcpyDeviceToHost) != hipSuccess)
        rc = AISX_ERR_HIP;
    if (rc == AISX_OK && hipDeviceSynchronize() != hipSuccess)
        rc = AISX_ERR_HIP;
    return rc == AISX_OK ? nout : rc;
}

// freqest::work(noutput_items, input_items, output_items) as the scheduler calls it
// (lib/freqest_impl.h:39-41, lib/freqest_impl.cc:57-88): input_items[0] = noutput_items vectors of
// fftlen gr_complex (item size 8 * fftlen, :43), output_items[0] = one float per vector (:44).
extern "C" int aisx_freqest_work_host(aisx_freqsync* h, int noutput_items, const aisx_cf32* in, float* out)
{
    if (!h || !in || !out || noutput_items < 0)
        return AISX_ERR_INVALID;
    if (h->nchan != 1) {
        set_err("aisx_freqest_work_host: handle has %d channels, the GNU Radio path needs 1", h->nchan);
        return AISX_ERR_INVALID;
    }
    if (noutput_items == 0)
        return 0;
    int rc;
    if (noutput_items > h->st_cap) {
        dev_free(h->d_st_vec);
        dev_free(h->d_st_out);
        h->st_cap = 0;
        if ((rc = dev_alloc(&h->d_st_vec, (size_t)noutput_items * h->fftlen, false)) != AISX_OK ||
            (rc = dev_alloc(&h->d_st_out, noutput_items, false)) != AISX_OK)
            return rc;
        h->st_cap = noutput_items;
    }
    const size_t nitems = (size_t)noutput_items * h->fftlen;
    AISX_HIPCHK(hipMemcpy(h->d_st_vec, in, sizeof(cf) * nitems, hipMemcpyHostToDevice));
    if ((rc = aisx_freqest_work(h, (const aisx_cf32*)h->d_st_vec, (long)nitems, h->d_st_out, noutput_items, noutput_items,
                                nullptr)) != AISX_OK)
        return rc;
    AISX_HIPCHK(hipMemcpy(out, h->d_st_out, sizeof(float) * noutput_items, hipMemcpyDeviceToHost));
    return noutput_items; // :87 return noutput_items
}

// ---------------------------------------------------------------------------
struct aisx_agc {
    int nchan = 0, W = 0, max_items = 0;
    float reference = 0, floor_env = AGC_FLOOR_DEFAULT;
    cf* d_hist[2] = { nullptr, nullptr };
    int cur = 0;
    bool tiles_only = false; // aisx_agc_set_streaming(h, 0): the tile kernels for every call
    int lds_claim = 0; // aisx_agc_set_lds_claim: LDS a streaming workgroup claims beyond the 8 KB it uses
    int lds_attr = 64 * 1024; // dynamic LDS a k_agcw<true> launch of THIS handle may ask for (raised per handle: handles
                              // may live on different devices, and the attribute is a per-device one)
    cf *d_hst_in = nullptr, *d_hst_out = nullptr; // aisx_agc_work_host's staging (grown, never shrunk)
    size_t hst_in_cap = 0, hst_out_cap = 0;
};

extern "C" int aisx_agc_geometry(const aisx_agc* h, int* nchan, int* max_items, int* nsamples, int* fused_ok)
{
    if (h && fused_ok)
        *fused_ok = agc8_applies(h->W) ? 1 : 0;
    if (!h)
        return AISX_ERR_INVALID;
    if (nchan)
        *nchan = h->nchan;
    if (max_items)
        *max_items = h->max_items;
    if (nsamples)
        *nsamples = h->W;
    return AISX_OK;
}

extern "C" int aisx_agc_create(aisx_agc** out, int nsamples, float reference, int nchan, int max_items)
{
    if (!out)
        return AISX_ERR_INVALID;
    *out = nullptr;
    if (nsamples < 1) { // [GR] feedforward_agc_cc: std::invalid_argument
        set_err("feedforward_agc_cc_impl: nsamples must be >= 1");
        return AISX_ERR_INVALID;
    }
    if (nsamples > AGC_MAXW || nchan < 1 || max_items < 1) {
        set_err("aisx_agc_create: bad argument (window %d, supported up to %d)", nsamples, AGC_MAXW);
        return AISX_ERR_INVALID;
    }
    int rc = require_device();
    if (rc != AISX_OK)
        return rc;
    aisx_agc* h = new aisx_agc();
    h->nchan = nchan;
    h->W = nsamples;
    h->max_items = max_items;
    h->reference = reference;
    if (const char* e = exp_env("AISX_AGC_STREAMING")) // (experiments; the API is aisx_agc_set_streaming)
        h->tiles_only = atoi(e) == 0;
    if ((rc = dev_alloc(&h->d_hist[0], (size_t)nchan * nsamples)) != AISX_OK ||
        (rc = dev_alloc(&h->d_hist[1], (size_t)nchan * nsamples)) != AISX_OK) {
        aisx_agc_destroy(h);
        return rc;
    }
    // dev_alloc's zero fill runs on the null stream; callers launch on their own (non-blocking)
    // streams, which do not order against it
    if (hipDeviceSynchronize() != hipSuccess) {
        set_err("aisx_agc_create: device synchronisation failed");
        aisx_agc_destroy(h);
        return AISX_ERR_HIP;
    }
    *out = h;
    return AISX_OK;
}

extern "C" int aisx_agc_set_floor(aisx_agc* h, float floor_env)
{
    if (!h || !(floor_env > 0.f) || !(floor_env <= 3.4028234663852886e38f)) { // (NaN and +inf fail the second test)
        set_err("aisx_agc_set_floor: the floor must be positive and finite");
        return AISX_ERR_INVALID;
    }
    h->floor_env = floor_env;
    return AISX_OK;
}

extern "C" int aisx_freqsync_set_walk_lds_claim(aisx_freqsync* h, int bytes)
{
    if (!h || bytes < 0 || bytes > 144 * 1024) {
        set_err("aisx_freqsync_set_walk_lds_claim: 0 .. 147456 bytes");
        return AISX_ERR_INVALID;
    }
    h->walk_claim = bytes;
    return AISX_OK;
}

extern "C" int aisx_freqsync_get_walk_lds_claim(const aisx_freqsync* h, int* bytes, int* used_bytes)
{
    if (!h)
        return AISX_ERR_INVALID;
    if (bytes)
        *bytes = h->walk_claim;
    if (used_bytes)
        *used_bytes = FSW_LDS_BYTES;
    return AISX_OK;
}

extern "C" int aisx_agc_set_lds_claim(aisx_agc* h, int bytes)
{
    if (!h || bytes < 0 || bytes > 144 * 1024) {
        set_err("aisx_agc_set_lds_claim: 0 .. 147456 bytes");
        return AISX_ERR_INVALID;
    }
    h->lds_claim = bytes;
    return AISX_OK;
}

extern "C" int aisx_agc_get_lds_claim(const aisx_agc* h, int* bytes, int* used_bytes)
{
    if (!h)
        return AISX_ERR_INVALID;
    if (bytes)
        *bytes = h->lds_claim;
    if (used_bytes)
        *used_bytes = AGW_LDS_BYTES;
    return AISX_OK;
}

extern "C" int aisx_agc_set_streaming(aisx_agc* h, int on)
{
    if (!h)
        return AISX_ERR_INVALID;
    h->tiles_only = !on;
    return AISX_OK;
}

extern "C" int aisx_agc_destroy(aisx_agc* h)
{
    if (!h)
        return AISX_OK;
    dev_free(h->d_hist[0]);
    dev_free(h->d_hist[1]);
    dev_free(h->d_hst_in);
    dev_free(h->d_hst_out);
    delete h;
    return AISX_OK;
}

extern "C" int aisx_agc_reset(aisx_agc* h)
{
    if (!h)
        return AISX_ERR_INVALID;
    AISX_HIPCHK(hipMemset(h->d_hist[0], 0, sizeof(cf) * (size_t)h->nchan * h->W));
    AISX_HIPCHK(hipMemset(h->d_hist[1], 0, sizeof(cf) * (size_t)h->nchan * h->W));
    AISX_HIPCHK(hipDeviceSynchronize());
    h->cur = 0;
    return AISX_OK;
}

extern "C" int aisx_agc_process(aisx_agc* h, const aisx_cf32* d_in, long in_stride, aisx_cf32* d_out, long out_stride,
                                int n, void* stream)
{
    if (!h || !d_in || !d_out || n < 1 || n > h->max_items || in_stride < n || out_stride < n) {
        set_err("aisx_agc_process: bad argument");
        return AISX_ERR_INVALID;
    }
    AgcParams p;
    p.in = (const cf*)d_in;
    p.in_stride = in_stride;
    p.out = (cf*)d_out;
    p.out_stride = out_stride;
    p.hist_in = h->d_hist[h->cur];
    p.hist_out = h->d_hist[h->cur ^ 1];
    p.n = n;
    p.W = h->W;
    p.reference = h->reference;
    p.floor_env = h->floor_env;
    const int TL = agc8_applies(h->W) ? AGC8_TL : AGC_TL;
    p.ntiles = (n + TL - 1) / TL;
    p.phases = nullptr;
    p.phases_stride = 0;
    p.dvec = nullptr;
    p.dvec_stride = 0;
    p.sintab = nullptr;
    p.pend_in = nullptr;
    p.pend_out = nullptr;
    p.npend = 0;
    p.n_raw = 0;
    if (agcw_applies(p.W, n) && !h->tiles_only)
        hipLaunchKernelGGL(k_agcw<false>, dim3(agcw_grid(n), h->nchan), dim3(AGW_T), 0, (hipStream_t)stream, p);
    else if (agc8_applies(p.W))
        hipLaunchKernelGGL(k_agc8, dim3(p.ntiles, h->nchan), dim3(AGC8_T), AGC8_LDS_BYTES, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(k_agc, dim3(p.ntiles, h->nchan), dim3(AGC_T), AGC_LDS_BYTES, (hipStream_t)stream, p);
    AISX_HIPCHK(hipGetLastError());
    h->cur ^= 1;
    return AISX_OK;
}

// ---- fused front end ------------------------------------------------------------------------
static int fs_fused_prepare(aisx_freqsync* h)
{
    int rc;
    if (!h->ev_walk) {
        AISX_HIPCHK(hipEventCreateWithFlags(&h->ev_walk, hipEventDisableTiming));
        AISX_HIPCHK(hipEventCreateWithFlags(&h->ev_proc, hipEventDisableTiming));
        AISX_HIPCHK(hipEventCreateWithFlags(&h->ev_est, hipEventDisableTiming));
    }
    for (int k = 0; k < 2; k++) {
        aisx_freqsync::Slot& s = h->slot[k];
        if (!s.ev_read)
            AISX_HIPCHK(hipEventCreateWithFlags(&s.ev_read, hipEventDisableTiming));
        if (!s.ev_ready)
            AISX_HIPCHK(hipEventCreateWithFlags(&s.ev_ready, hipEventDisableTiming));
        if (!s.d_maxpos && (rc = dev_alloc(&s.d_maxpos, (size_t)h->nchan * h->max_vec)) != AISX_OK)
            return rc;
        if (!s.d_fhat && (rc = dev_alloc(&s.d_fhat, (size_t)h->nchan * h->max_vec)) != AISX_OK)
            return rc;
        // (each allocation under its own test: a call that failed half way is finished by the next one)
        if (!s.d_dvec && (rc = dev_alloc(&s.d_dvec, (size_t)h->nchan * h->max_vec)) != AISX_OK)
            return rc;
        if (!s.d_phases) {
            h->phases_stride = ((long)h->max_vec * (h->fftlen / FSW_CK) + 3) & ~3L;
            if ((rc = dev_alloc(&s.d_phases, (size_t)h->nchan * (size_t)h->phases_stride, false)) != AISX_OK)
                return rc;
        }
    }
    return AISX_OK;
}

// frequency estimates (fs_est_body) and NCO phase walk (fs_walk_body) of the call that comes
// next (depth 0) or of the one after it (depth 1: only behind a call that leaves no pending items),
// into the slot it will use; the walk leaves the phase it ends on in an uncommitted copy
static int fs_estimate_into_slot(aisx_freqsync* h, const aisx_cf32* d_in, long in_stride, int n, hipStream_t st,
                                 hipStream_t st_walk, int depth = 0)
{
    const int npend = depth ? 0 : h->npend;
    const int nvec = (npend + n) / h->fftlen;
    if (nvec == 0)
        return AISX_OK;
    aisx_freqsync::Slot& s = h->slot[h->slot_cur ^ depth];
    // what this estimate reads or overwrites may still be in use on another stream
    if (s.read_pending)
        AISX_HIPCHK(hipStreamWaitEvent(st, s.ev_read, 0)); // the pass of two calls ago read this slot
    if (h->proc_pending && npend > 0)
        AISX_HIPCHK(hipStreamWaitEvent(st, h->ev_proc, 0)); // the last pass wrote the pending items
    if (h->walk_pending)
        AISX_HIPCHK(hipStreamWaitEvent(st_walk, h->ev_walk, 0)); // the last walk wrote the phase this one starts from
    FsEstParams e;
    e.in = (const cf*)d_in;
    e.in_stride = in_stride;
    e.pend = h->d_pend[h->cur];
    e.npend = npend;
    e.wtab = h->d_wtab;
    e.maxpos = s.d_maxpos;
    e.maxpos_stride = h->max_vec;
    e.nvec = nvec;
    e.offset = h->offset;
    static const int est_pad = exp_env("AISX_EST_LDS_PAD") ? atoi(exp_env("AISX_EST_LDS_PAD")) : 0; // (experiments: placement)
    hipLaunchKernelGGL(k_fs_est, dim3((nvec + FS_VEC_PER_WG - 1) / FS_VEC_PER_WG, h->nchan), dim3(FS_T), FS_LDS_BYTES + est_pad, st, e);
    AISX_HIPCHK(hipGetLastError());
    if (st_walk != st) { // the walk on a stream of its own, behind the estimates
        AISX_HIPCHK(hipEventRecord(h->ev_est, st));
        AISX_HIPCHK(hipStreamWaitEvent(st_walk, h->ev_est, 0));
        if (s.read_pending)
            AISX_HIPCHK(hipStreamWaitEvent(st_walk, s.ev_read, 0));
    }
    FsWalkParams w;
    w.nchan = h->nchan;
    w.maxpos = s.d_maxpos;
    w.maxpos_stride = h->max_vec;
    w.fhat = s.d_fhat;
    w.fhat_stride = h->max_vec;
    w.phase_in = h->d_phase3[(h->phase_cur + depth) % 3];
    w.phase_out = h->d_phase3[(h->phase_cur + depth + 1) % 3];
    w.phases = s.d_phases;
    w.phases_stride = h->phases_stride;
    w.dvec = s.d_dvec;
    w.dvec_stride = h->max_vec;
    w.nvec = nvec;
    w.binsize = h->binsize;
    w.sensitivity = h->sensitivity;
    // (LDS the walk's one-wave workgroups claim beyond what they use: aisx_freqsync_set_walk_lds_claim; experiments:
    // AISX_WALK_LDS_PAD, bytes, overrides it)
    static const int walk_pad_env = exp_env("AISX_WALK_LDS_PAD") ? atoi(exp_env("AISX_WALK_LDS_PAD")) : -1;
    const int walk_pad = walk_pad_env >= 0 ? walk_pad_env : h->walk_claim;
    if (FSW_LDS_BYTES + walk_pad > h->walk_attr) {
        AISX_HIPCHK(hipFuncSetAttribute((const void*)k_fs_walk, hipFuncAttributeMaxDynamicSharedMemorySize, FSW_LDS_BYTES + walk_pad));
        h->walk_attr = FSW_LDS_BYTES + walk_pad;
    }
    hipLaunchKernelGGL(k_fs_walk, dim3((h->nchan + FSW_T - 1) / FSW_T), dim3(FSW_T), FSW_LDS_BYTES + walk_pad, st_walk, w);
    AISX_HIPCHK(hipGetLastError());
    AISX_HIPCHK(hipEventRecord(h->ev_walk, st_walk));
    AISX_HIPCHK(hipEventRecord(s.ev_ready, st_walk));
    h->walk_pending = true;
    return AISX_OK;
}

// Prepares, on `stream`, the frequency estimates and NCO phases of the NEXT
// aisx_freqsync_agc_process call, which must come with the same d_in / in_stride / n (otherwise
// the preparation is dropped and that call estimates for itself).  The serial phase walk of call
// k + 1 can so run beside the sample passes of call k.  A second estimate (for call k + 2) may be
// prepared while the first is still waiting, provided call k + 1 leaves no pending items (nothing
// pending now, its n a multiple of fftlen): a caller that issues estimate_ahead(k + 1) BEFORE
// process(k) gives the walk the whole of step k to hide in.
extern "C" int aisx_freqsync_estimate_ahead(aisx_freqsync* h, const aisx_cf32* d_in, long in_stride, int n, void* stream,
                                            void* walk_stream)
{
    if (fs_whole(h, "aisx_freqsync_estimate_ahead") != AISX_OK)
        return AISX_ERR_INVALID;
    if (!h || !d_in || n < 1 || n > h->max_items || in_stride < n) {
        set_err("aisx_freqsync_estimate_ahead: bad argument");
        return AISX_ERR_INVALID;
    }
    if (h->ahead_cnt == 2) {
        set_err("aisx_freqsync_estimate_ahead: two estimates are already waiting for their aisx_freqsync_agc_process calls");
        return AISX_ERR_INVALID;
    }
    if (h->ahead_cnt == 1 && (h->npend != 0 || h->ahead_q[0].n % h->fftlen != 0)) {
        set_err("aisx_freqsync_estimate_ahead: a second estimate can be prepared only behind a call that leaves no pending "
                "items (%d pending now, n = %d of the call waiting, fftlen %d)", h->npend, h->ahead_q[0].n, h->fftlen);
        return AISX_ERR_INVALID;
    }
    int rc;
    hipStream_t sw = walk_stream ? (hipStream_t)walk_stream : (hipStream_t)stream;
    const int depth = h->ahead_cnt;
    if ((rc = fs_fused_prepare(h)) != AISX_OK ||
        (rc = fs_estimate_into_slot(h, d_in, in_stride, n, (hipStream_t)stream, sw, depth)) != AISX_OK)
        return rc;
    h->ahead_q[depth].in = d_in;
    h->ahead_q[depth].stride = in_stride;
    h->ahead_q[depth].n = n;
    h->ahead_cnt = depth + 1;
    return AISX_OK;
}

// The first two blocks of python/ais_demod.py:56 in one pass over the samples: the frequency
// estimates (fs_est_body) and the NCO phase walk (fs_walk_body) as in aisx_freqsync_process, the
// mixing done where feedforward_agc_cc reads its input (agc8_body): square_and_fft_sync_cc's
// output is never stored.  Results are those of aisx_freqsync_process followed by
// aisx_agc_process on its output, bit for bit; both handles advance as if those had been called.
extern "C" int aisx_freqsync_agc_process(aisx_freqsync* h, aisx_agc* a, const aisx_cf32* d_in, long in_stride, int n,
                                         aisx_cf32* d_out, long out_stride, float* d_fhat, long fhat_stride, int* n_out,
                                         void* stream)
{
    if (fs_whole(h, "aisx_freqsync_agc_process") != AISX_OK)
        return AISX_ERR_INVALID;
    if (!h || !a || !d_in || !d_out || !n_out || n < 1 || n > h->max_items || in_stride < n || a->nchan != h->nchan) {
        set_err("aisx_freqsync_agc_process: bad argument");
        return AISX_ERR_INVALID;
    }
    if (!agc8_applies(a->W)) {
        set_err("aisx_freqsync_agc_process: the fused front end serves AGC windows that are a multiple of 8 in [16, %d] "
                "(the stock 512 is); use aisx_freqsync_process + aisx_agc_process for %d", AGC_MAXW, a->W);
        return AISX_ERR_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nvec = (h->npend + n) / h->fftlen;
    const int total = nvec * h->fftlen;
    if (total > a->max_items || out_stride < total || (d_fhat && fhat_stride < nvec)) {
        set_err("aisx_freqsync_agc_process: %d output items exceed the AGC's max_items %d or the output stride", total, a->max_items);
        return AISX_ERR_INVALID;
    }
    int rc;
    if ((rc = fs_fused_prepare(h)) != AISX_OK)
        return rc;
    if (h->ahead_cnt > 0 && (h->ahead_q[0].in != (const void*)d_in || h->ahead_q[0].stride != in_stride || h->ahead_q[0].n != n)) {
        // prepared for other arguments: nothing of it was committed, estimate afresh -- behind whatever
        // the dropped preparations still have running (they write the slots and phase copies used next)
        h->ahead_cnt = 0;
        if (h->walk_pending)
            AISX_HIPCHK(hipStreamWaitEvent(st, h->ev_walk, 0));
    }
    aisx_freqsync::Slot& s = h->slot[h->slot_cur];
    if (h->ahead_cnt > 0) {
        if (nvec > 0)
            AISX_HIPCHK(hipStreamWaitEvent(st, s.ev_ready, 0));
        h->ahead_q[0] = h->ahead_q[1];
        h->ahead_cnt--;
    } else if ((rc = fs_estimate_into_slot(h, d_in, in_stride, n, st, st)) != AISX_OK) {
        return rc;
    }
    if (nvec > 0) {
        h->phase_cur = (h->phase_cur + 1) % 3; // the walk's end phase becomes the block's d_phase
        h->d_phase = h->d_phase3[h->phase_cur];
        if (d_fhat)
            AISX_HIPCHK(hipMemcpy2DAsync(d_fhat, sizeof(float) * fhat_stride, s.d_fhat, sizeof(float) * h->max_vec, sizeof(float) * nvec,
                                         h->nchan, hipMemcpyDeviceToDevice, st));
    }
    AgcParams p;
    p.in = (const cf*)d_in;
    p.in_stride = in_stride;
    p.out = (cf*)d_out;
    p.out_stride = out_stride;
    p.hist_in = a->d_hist[a->cur];
    p.hist_out = a->d_hist[a->cur ^ 1];
    p.n = total;
    p.W = a->W;
    p.reference = a->reference;
    p.floor_env = a->floor_env;
    p.ntiles = total > 0 ? (total + AGC8_TL - 1) / AGC8_TL : 1; // (a call without a whole vector still moves the pending items)
    p.phases = s.d_phases;
    p.phases_stride = h->phases_stride;
    p.dvec = s.d_dvec;
    p.dvec_stride = h->max_vec;
    p.sintab = h->d_sintab;
    p.pend_in = h->d_pend[h->cur];
    p.pend_out = h->d_pend[h->cur ^ 1];
    p.npend = h->npend;
    p.n_raw = n;
    static const int agcw_pad = exp_env("AISX_AGCW_LDS_PAD") ? atoi(exp_env("AISX_AGCW_LDS_PAD")) : -1; // (experiments: overrides the handle's claim)
    if (agcw_applies(p.W, total) && !a->tiles_only) {
        const int lds = AGW_LDS_BYTES + (agcw_pad >= 0 ? agcw_pad : a->lds_claim);
        if (lds > a->lds_attr) { // (what a launch may ask for before the kernel's limit is raised)
            AISX_HIPCHK(hipFuncSetAttribute((const void*)k_agcw<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            a->lds_attr = lds;
        }
        hipLaunchKernelGGL(k_agcw<true>, dim3(agcw_grid(total), h->nchan), dim3(AGW_T), lds, st, p);
    }
    else
        hipLaunchKernelGGL(k_agc8, dim3(p.ntiles, h->nchan), dim3(AGC8_T), AGC8_LDS_BYTES_MIXED, st, p);
    AISX_HIPCHK(hipGetLastError());
    AISX_HIPCHK(hipEventRecord(s.ev_read, st));
    s.read_pending = true;
    AISX_HIPCHK(hipEventRecord(h->ev_proc, st));
    h->proc_pending = true;
    h->slot_cur ^= 1;
    h->npend = h->npend + n - total;
    h->cur ^= 1;
    a->cur ^= 1;
    *n_out = total;
    return AISX_OK;
}

// feedforward_agc_cc::work as the scheduler calls it ([GR] sync_block with set_history(nsamples)):
// `in` = input_items[0] = nsamples - 1 old items followed by noutput_items new ones, host pointers,
// nchan == 1.  Returns noutput_items.
extern "C" int aisx_agc_work_host(aisx_agc* h, int noutput_items, const aisx_cf32* in, aisx_cf32* out)
{
    if (!h || !in || !out || noutput_items < 1 || noutput_items > h->max_items)
        return AISX_ERR_INVALID;
    if (h->nchan != 1) {
        set_err("aisx_agc_work_host: handle has %d channels, the GNU Radio path needs 1", h->nchan);
        return AISX_ERR_INVALID;
    }
    const int H = h->W - 1, n = noutput_items;
    int rc = stage_grow(&h->d_hst_in, &h->hst_in_cap, (size_t)n);
    if (rc == AISX_OK)
        rc = stage_grow(&h->d_hst_out, &h->hst_out_cap, (size_t)n);
    cf *d_in = h->d_hst_in, *d_out = h->d_hst_out;
    // the block's history comes from the scheduler's buffer, not from the handle
    if (rc == AISX_OK && H > 0 && hipMemcpy(h->d_hist[h->cur], in, sizeof(cf) * H, hipMemcpyHostToDevice) != hipSuccess)
        rc = AISX_ERR_HIP;
    if (rc == AISX_OK && hipMemcpy(d_in, in + H, sizeof(cf) * n, hipMemcpyHostToDevice) != hipSuccess)
        rc = AISX_ERR_HIP;
    if (rc == AISX_OK)
        rc = aisx_agc_process(h, (const aisx_cf32*)d_in, n, (aisx_cf32*)d_out, n, n, nullptr);
    if (rc == AISX_OK && hipMemcpy(out, d_out, sizeof(cf) * n, hipMemcpyDeviceToHost) != hipSuccess)
        rc = AISX_ERR_HIP;
    return rc == AISX_OK ? n : rc;
}

// ---------------------------------------------------------------------------
// N3: polyphase channelizer front end (BASELINE config 5)
// ---------------------------------------------------------------------------
#include "k_pfb.h"

__global__ __launch_bounds__(PFB_T) void k_pfb(PfbParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
    pfb_body(cx, p);
}

struct aisx_pfb {
    int nstreams = 0, D = 0, K = 0, Lh = 0, max_frames = 0;
    float* d_taps = nullptr;
    cf* d_wtab = nullptr;
    cf* d_hist[2] = { nullptr, nullptr };
    int cur = 0;
    long frame0 = 0;
};

extern "C" int aisx_pfb_create(aisx_pfb** out, int nlanes, int decim, const float* taps, int ntaps, int nstreams,
                               int max_frames)
{
    if (!out)
        return AISX_ERR_INVALID;
    *out = nullptr;
    if (nlanes != PFB_M || (decim != PFB_M && decim != PFB_M / 2) || !taps || ntaps < 1 || nstreams < 1 || max_frames < 1) {
        set_err("aisx_pfb_create: supported geometry is %d lanes, decimation %d or %d", PFB_M, PFB_M, PFB_M / 2);
        return AISX_ERR_INVALID;
    }
    int rc = require_device();
    if (rc != AISX_OK)
        return rc;
    aisx_pfb* h = new aisx_pfb();
    h->nstreams = nstreams;
    h->D = decim;
    h->K = (ntaps + PFB_M - 1) / PFB_M;
    h->Lh = h->K * PFB_M;
    h->max_frames = max_frames;
    std::vector<float> pad((size_t)h->Lh, 0.f);
    for (int i = 0; i < ntaps; i++)
        pad[i] = taps[i];
    std::vector<cf> w(PFB_M);
    for (int k = 0; k < PFB_M; k++) {
        double a = -2.0 * M_PI * (double)k / (double)PFB_M;
        w[k] = mk((float)cos(a), (float)sin(a));
    }
    if ((rc = dev_alloc(&h->d_taps, h->Lh)) != AISX_OK || (rc = dev_alloc(&h->d_wtab, PFB_M)) != AISX_OK ||
        (rc = dev_alloc(&h->d_hist[0], (size_t)nstreams * h->Lh)) != AISX_OK ||
        (rc = dev_alloc(&h->d_hist[1], (size_t)nstreams * h->Lh)) != AISX_OK) {
        aisx_pfb_destroy(h);
        return rc;
    }
    if (hipMemcpy(h->d_taps, pad.data(), sizeof(float) * h->Lh, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->d_wtab, w.data(), sizeof(cf) * PFB_M, hipMemcpyHostToDevice) != hipSuccess) {
        set_err("aisx_pfb_create: table upload failed");
        aisx_pfb_destroy(h);
        return AISX_ERR_HIP;
    }
    *out = h;
    return AISX_OK;
}

extern "C" int aisx_pfb_destroy(aisx_pfb* h)
{
    if (!h)
        return AISX_OK;
    dev_free(h->d_taps);
    dev_free(h->d_wtab);
    dev_free(h->d_hist[0]);
    dev_free(h->d_hist[1]);
    delete h;
    return AISX_OK;
}

extern "C" int aisx_pfb_process(aisx_pfb* h, const aisx_cf32* d_in, long in_stride, int n, aisx_cf32* d_out,
                                long out_stride, int* nframes, void* stream)
{
    if (!h || !d_in || !d_out || !nframes || n < 1 || (n % h->D) != 0 || in_stride < n) {
        set_err("aisx_pfb_process: n must be a positive multiple of the decimation %d", h ? h->D : 0);
        return AISX_ERR_INVALID;
    }
    const int nf = n / h->D;
    if (nf > h->max_frames || out_stride < nf) {
        set_err("aisx_pfb_process: %d frames exceed the capacity", nf);
        return AISX_ERR_INVALID;
    }
    PfbParams p;
    p.in = (const cf*)d_in;
    p.in_stride = in_stride;
    p.hist_in = h->d_hist[h->cur];
    p.hist_out = h->d_hist[h->cur ^ 1];
    p.taps = h->d_taps;
    p.wtab = h->d_wtab;
    p.out = (cf*)d_out;
    p.out_stride = out_stride;
    p.n = n;
    p.D = h->D;
    p.K = h->K;
    p.Lh = h->Lh;
    p.nframes = nf;
    p.frame0 = h->frame0;
    hipLaunchKernelGGL(k_pfb, dim3((nf + 3) / 4, h->nstreams), dim3(PFB_T), PFB_LDS_BYTES, (hipStream_t)stream, p);
    AISX_HIPCHK(hipGetLastError());
    h->cur ^= 1;
    h->frame0 += nf;
    *nframes = nf;
    return AISX_OK;
}
