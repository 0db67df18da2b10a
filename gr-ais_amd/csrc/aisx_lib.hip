// aisx_lib.hip -- C ABI (include/aisx.h) for corr_est_cc and msk_timing_recovery_cc
// plus the __global__ wrappers that run the kernel bodies on gfx950.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "aisx_devctx.h"
#include "aisx_host.h"
#include "aisx_plan.h"
#include "aisx_tables.h"
#include "k_corr.h"
#include <cstdlib>

using namespace aisx;

namespace aisx {
char* err_buf()
{
    static thread_local char buf[512] = "";
    return buf;
}
} // namespace aisx

extern "C" int aisx_version(void) { return AISX_VERSION; }
extern "C" const char* aisx_last_error(void) { return err_buf(); }
extern "C" int aisx_device_count(int* count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (count)
        *count = (e == hipSuccess) ? n : 0;
    return e == hipSuccess ? AISX_OK : AISX_ERR_NO_DEVICE;
}
extern "C" int aisx_set_device(int device)
{
    AISX_HIPCHK(hipSetDevice(device));
    return AISX_OK;
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(CF_T) void k_corr_inith(CorrInitParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtxC cx{ { smem } };
    corr_inith_body(cx, p);
}

// The product launches two correlators: k_corr2d_main (templates up to 512 samples) and k_corr4f_main (513 .. 2048).
// Their predecessors -- round 1's k_corr_main / k_corr4_main (plain window loads), k_corr4d_main (256 threads x 16 points,
// LDS-DMA), k_corr4e_main (512 threads x 8 points, LDS-DMA) -- exist in the experiments build only (AISX_CORR_DMA=0,
// AISX_CORR_WIDE=0 / 1): A/B partners and the twins of tests/test_gpu_corr_msk.py.
#ifdef AISX_EXPERIMENTS
__global__ __launch_bounds__(CF_T, 3) void k_corr_main(CorrParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtxC cx{ { smem } };
    corr_main_body(cx, p);
}

__global__ __launch_bounds__(CF4_T) void k_corr4_inith(CorrInitParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
    corr4_inith_body(cx, p);
}

__global__ __launch_bounds__(CF4_T, 3) void k_corr4_main(CorrParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
    corr4_main_body(cx, p);
}

// the F = 4096 correlator with the next tile's window prefetched by LDS-DMA (k_corr4d.h): two
// workgroups per CU (two 34 KB window images each), up to 256 VGPRs
// (Two waves per SIMD at 256 VGPRs leave no register room for a wave of another stream's kernel on
// that SIMD: beside the NCO phase walk (30 VGPRs, one wave on each of 64 CUs) or the timing
// recovery (105) a CU holds one workgroup of this kernel, not two.  amdgpu_num_vgpr(240) does not
// cap the unified file -- the compiler spills into AGPRs on top -- so this stays as it is.)
template <int NC>
__global__ __launch_bounds__(CF4_T, 2) void k_corr4d_main(CorrParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
#ifdef CORR_PRIO
    __builtin_amdgcn_s_setprio(CORR_PRIO);
#endif
    corr4d_main_body<DevCtx, NC>(cx, p);
}
// builds with the template length folded in: the stock template at 4 and at 5 samples per symbol
// (224 symbols, python/ais_demod.py:36-38) and config 5's two lengths; every other length runs the <0> build
static void (*corr4d_pick(int N))(CorrParams)
{
    switch (N) {
    case 896: return k_corr4d_main<896>;
    case 1120: return k_corr4d_main<1120>;
    // the lengths BASELINE config 5 runs at: the wideband benchmark's template cut to 1024 items (bench.py --chain
    // wideband) and the 224-symbol template at the channelizer's 48 828.125 Hz lane rate (1139 items, tests/test_gpu_configs.py)
    case 1024: return k_corr4d_main<1024>;
    case 1139: return k_corr4d_main<1139>;
    default: return k_corr4d_main<0>; // any other length: the run-time-length build (no scratch either, k_corr4d.h)
    }
}

#ifdef CE_PROF
namespace aisx {
__device__ unsigned long long g_ce_prof[16];
}
extern "C" int aisx_debug_ce_prof(unsigned long long* out16, int reset)
{
    AISX_HIPCHK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(aisx::g_ce_prof), 16 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[16] = {};
        AISX_HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(aisx::g_ce_prof), z, sizeof(z)));
    }
    return AISX_OK;
}
#endif
#endif // AISX_EXPERIMENTS
// the F = 4096 correlator on 512 threads x 8 points (k_corr4e.h: the plan; k_corr4f.h: the kernel the product runs):
// two workgroups per CU, at most 128 VGPRs: four waves per SIMD
__global__ __launch_bounds__(CE_T) void k_corr4e_inith(CorrInitParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
    corr4e_inith_body(cx, p);
}
#ifdef AISX_EXPERIMENTS
template <int NC>
__global__ __launch_bounds__(CE_T, 4) void k_corr4e_main(CorrParams p) // (second argument: waves per SIMD)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
    corr4e_main_body<DevCtx, NC>(cx, p);
}
#endif
// ... and with the next window fetched into registers (k_corr4f.h): one image + a small overlap buffer
template <int NC>
__global__ __launch_bounds__(CE_T, 4) void k_corr4f_main(CorrParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
    corr4f_main_body<DevCtx, NC>(cx, p);
}
static void (*corr4f_pick(int N))(CorrParams)
{
    switch (N) {
    case 896: return k_corr4f_main<896>;
    case 112: return k_corr4f_main<112>;
    case 1120: return k_corr4f_main<1120>;
    case 1024: return k_corr4f_main<1024>;
    case 1139: return k_corr4f_main<1139>;
    default: return k_corr4f_main<0>;
    }
}
#ifdef AISX_EXPERIMENTS
static void (*corr4e_pick(int N))(CorrParams)
{
    switch (N) {
    case 896: return k_corr4e_main<896>;
    default: return k_corr4e_main<0>;
    }
}
#endif

// the F = 2048 correlator with the next tile's window prefetched by LDS-DMA (k_corr2d.h): four
// workgroups of two waves per CU (two 17 KB window images each), up to 256 VGPRs
template <int NC>
__global__ __launch_bounds__(CF_T, 2) void k_corr2d_main(CorrParams p) // (second argument: waves per SIMD)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevCtx cx{ smem };
    corr2d_main_body<DevCtx, NC>(cx, p);
}
// builds with the template length folded in: the 28-symbol preamble at 4 and at 5 samples per symbol
static void (*corr2d_pick(int N))(CorrParams)
{
    switch (N) {
    case 112: return k_corr2d_main<112>;
    case 140: return k_corr2d_main<140>;
    default: return k_corr2d_main<0>;
    }
}

__global__ __launch_bounds__(64 * RSV_WAVES) void k_corr_resolve(ResolveParams p)
{
    // fast_atan2f's 257-entry table, the regions' detection counts, the waves' detections (k_corr.h: rsv_lds_bytes)
    extern __shared__ __attribute__((aligned(16))) char rsv[];
    DevCtx cx{ rsv };
    corr_resolve_body(cx, p);
}

// ---------------------------------------------------------------------------
// measurement hook: what a plain copy sustains on this chip (16 bytes per lane and access,
// grid-stride, no profiler attached) -- the practical ceiling next to the 8 TB/s spec peak the
// correlator's roofline fraction is quoted against
// ---------------------------------------------------------------------------
// MODE 0: one 16-byte access per lane and grid-stride step; 1: four independent ones in flight
// per lane; 2: four, non-temporal (streaming: no reuse, as the correlator's traffic)
typedef float copy_v4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k_copy16(const copy_v4* __restrict__ src, copy_v4* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (MODE >= 1) {
        for (; i + 3 * stride < n; i += 4 * stride) {
            copy_v4 a, b, c, d;
            if (MODE == 2) {
                a = __builtin_nontemporal_load(src + i);
                b = __builtin_nontemporal_load(src + i + stride);
                c = __builtin_nontemporal_load(src + i + 2 * stride);
                d = __builtin_nontemporal_load(src + i + 3 * stride);
                __builtin_nontemporal_store(a, dst + i);
                __builtin_nontemporal_store(b, dst + i + stride);
                __builtin_nontemporal_store(c, dst + i + 2 * stride);
                __builtin_nontemporal_store(d, dst + i + 3 * stride);
            } else {
                a = src[i];
                b = src[i + stride];
                c = src[i + 2 * stride];
                d = src[i + 3 * stride];
                dst[i] = a;
                dst[i + stride] = b;
                dst[i + 2 * stride] = c;
                dst[i + 3 * stride] = d;
            }
        }
    }
    for (; i < n; i += stride)
        dst[i] = src[i];
}

extern "C" int aisx_util_copy_GBs(size_t bytes, int iters, float* GBs)
{
    if (!GBs || bytes < (1u << 20) || iters < 1)
        return AISX_ERR_INVALID;
    int rc = require_device();
    if (rc != AISX_OK)
        return rc;
    const size_t n = bytes / 16;
    copy_v4 *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto done = [&](int r) {
        dev_free(a);
        dev_free(b);
        if (e0)
            (void)hipEventDestroy(e0);
        if (e1)
            (void)hipEventDestroy(e1);
        return r;
    };
    if ((rc = dev_alloc(&a, n)) != AISX_OK || (rc = dev_alloc(&b, n, false)) != AISX_OK)
        return done(rc);
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        return done(AISX_ERR_HIP);
    // several launch shapes; the best one is the figure (bytes read + bytes written per second)
    float best = 0.f;
    for (int mode = 0; mode < 3; mode++)
        for (int wg_per_cu : { 8, 16, 32 }) {
            const int grid = 256 * wg_per_cu;
            auto launch = [&] {
                if (mode == 0)
                    hipLaunchKernelGGL(k_copy16<0>, dim3(grid), dim3(256), 0, 0, a, b, n);
                else if (mode == 1)
                    hipLaunchKernelGGL(k_copy16<1>, dim3(grid), dim3(256), 0, 0, a, b, n);
                else
                    hipLaunchKernelGGL(k_copy16<2>, dim3(grid), dim3(256), 0, 0, a, b, n);
            };
            launch();
            if (hipEventRecord(e0, 0) != hipSuccess)
                return done(AISX_ERR_HIP);
            for (int k = 0; k < iters; k++)
                launch();
            float ms = 0.f;
            if (hipEventRecord(e1, 0) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f))
                return done(AISX_ERR_HIP);
            const float g = (float)(2.0 * (double)(n * 16) * iters / (ms * 1e-3) / 1e9);
            best = g > best ? g : best;
        }
    *GBs = best;
    return done(AISX_OK);
}

// ---------------------------------------------------------------------------
// corr_est_cc
// ---------------------------------------------------------------------------
// (experiments: AISX_CORR_F4=1 serves every template length with the F = 4096 build)
static int corr_pick_fft_x(int nsym)
{
    if (const char* e = exp_env("AISX_CORR_F4"))
        if (atoi(e) != 0)
            return CF4_F;
    return corr_pick_fft(nsym);
}

struct aisx_corr {
    int nchan = 0, N = 0, max_items = 0, tag_cap = 0, L = 0, isps = 0, out_multiple = 0;
    int F = CF_F; // FFT build serving this template length
    bool dma = true; // F = 4096: the k_corr4d.h build (AISX_CORR_DMA=0 selects k_corr4k.h's)
#ifndef AISX_CORR_WIDE_DEFAULT
#define AISX_CORR_WIDE_DEFAULT 2
#endif
    // F = 4096 with dma: 0 = k_corr4d.h, 1 = the 512-thread build k_corr4e.h, 2 = k_corr4f.h (1 and 2: the template
    // spectrum in the order of the 8 x 8 x 8 x 8 plan)
    int wide = AISX_CORR_WIDE_DEFAULT;
    const void* dma_attr_set = nullptr; // build whose dynamic-LDS limit has been raised ...
    int dma_attr_bytes = 0;             // ... and to how many bytes
    int lds_claim = 0; // aisx_corr_set_lds_claim: LDS a workgroup of the F = 4096 build claims beyond what it uses
    float sps = 0, thresh = 0;
    unsigned mark_delay = 0;
    std::vector<cf> symbols; // d_symbols
    cf *d_taps = nullptr, *d_tapspad = nullptr, *d_Hpos = nullptr, *d_wtab = nullptr;
    cf* d_hist[2] = { nullptr, nullptr };
    int hist_cur = 0;
    unsigned long long* d_abits = nullptr;
    long abits_stride = 0;
    cf* d_scratch = nullptr;
    long scratch_stride = 0;
    // tags rotate through three buffers so that a consumer on another stream (the
    // timing-recovery stage) can still read call k's tags while calls k+1 and k+2 run
    static constexpr int NTAGBUF = 3;
    tag_rec* d_tags2[NTAGBUF] = { nullptr, nullptr, nullptr };
    int* d_tag_count2[NTAGBUF] = { nullptr, nullptr, nullptr };
    int tag_cur = 0; // buffer the LAST call wrote
    tag_rec* d_tags = nullptr;
    int* d_tag_count = nullptr;
    float* d_atan = nullptr;
    uint64_t written = 0;
    int last_emit_port1 = 0;
    int corr_hist_zero = 0; // set by set_symbols(), consumed by the next call
    int prof = 0; // aisx_corr_set_profiling
    static constexpr int NEV = 64; // ring of event pairs: one per call, read back after the timed region
    hipEvent_t ev0[NEV] = {}, ev1[NEV] = {};
    long ncalls_prof = 0;
    // GNU Radio path staging
    cf *d_st_in = nullptr, *d_st_out = nullptr, *d_st_corr = nullptr;
    int st_cap = 0;
};

static int corr_upload_taps(aisx_corr* h)
{
    // taps/F, zero padded, then the forward transform in the kernel's own position order
    std::vector<cf> pad = corr_padded_taps(h->symbols, h->F);
    AISX_HIPCHK(hipMemcpy(h->d_tapspad, pad.data(), sizeof(cf) * h->F, hipMemcpyHostToDevice));
    AISX_HIPCHK(hipMemcpy(h->d_taps, h->symbols.data(), sizeof(cf) * h->N, hipMemcpyHostToDevice));
    CorrInitParams ip{ h->d_tapspad, h->d_wtab, h->d_Hpos };
    if (h->F == CF_F)
        hipLaunchKernelGGL(k_corr_inith, dim3(1), dim3(CF_T), CF_LDS_BYTES, 0, ip);
#ifdef AISX_EXPERIMENTS
    else if (!(h->dma && h->wide))
        hipLaunchKernelGGL(k_corr4_inith, dim3(1), dim3(CF4_T), CF4_LDS_BYTES, 0, ip);
#endif
    else {
        AISX_HIPCHK(hipFuncSetAttribute((const void*)k_corr4e_inith, hipFuncAttributeMaxDynamicSharedMemorySize, CE_LDS_BYTES));
        hipLaunchKernelGGL(k_corr4e_inith, dim3(1), dim3(CE_T), CE_LDS_BYTES, 0, ip);
    }
    AISX_HIPCHK(hipGetLastError());
    AISX_HIPCHK(hipDeviceSynchronize());
    return AISX_OK;
}

extern "C" int aisx_corr_create(aisx_corr** out, const aisx_cf32* symbols, int nsym, float sps, unsigned mark_delay,
                                float threshold, int nchan, int max_items, int max_tags_per_chan)
{
    if (!out)
        return AISX_ERR_INVALID;
    *out = nullptr;
    if (!symbols || nsym < 1 || nchan < 1 || max_items < 1 || max_tags_per_chan < 4) {
        set_err("aisx_corr_create: bad argument");
        return AISX_ERR_INVALID;
    }
    if (nsym > CORR_MAX_TEMPLATE) {
        set_err("aisx_corr_create: template of %d samples exceeds the %d supported (F = %d build)", nsym,
                CORR_MAX_TEMPLATE, CF4_F);
        return AISX_ERR_INVALID;
    }
    int rc = require_device();
    if (rc != AISX_OK)
        return rc;
    aisx_corr* h = new aisx_corr();
    h->nchan = nchan;
    h->N = nsym;
    h->max_items = max_items;
    h->tag_cap = max_tags_per_chan;
    h->sps = sps;
    h->F = corr_pick_fft_x(nsym);
    h->L = h->F - nsym;
    if (const char* e = exp_env("AISX_CORR_DMA")) // (experiments, and the tests of the other build)
        h->dma = atoi(e) != 0;
    if (const char* e = exp_env("AISX_CORR_WIDE"))
        h->wide = atoi(e);
    // constructor maths of lib/corr_est_cc_impl.cc:58-85 (aisx_plan.h)
    CorrSetup cs = corr_setup((const cf*)symbols, nsym, sps, mark_delay, threshold);
    h->symbols = cs.symbols;
    h->mark_delay = cs.mark_delay;
    h->thresh = cs.thresh;
    h->isps = cs.isps;
    h->out_multiple = cs.out_multiple;
    std::vector<cf> w = corr_wtab(h->F);
    h->abits_stride = (max_items + 63) / 64 + 1;
    h->scratch_stride = max_items;
#define CK(e)               \
    do {                    \
        rc = (e);           \
        if (rc != AISX_OK) { \
            aisx_corr_destroy(h); \
            return rc;      \
        }                   \
    } while (0)
    CK(dev_alloc(&h->d_taps, nsym));
    CK(dev_alloc(&h->d_tapspad, h->F));
    CK(dev_alloc(&h->d_Hpos, h->F));
    CK(dev_alloc(&h->d_wtab, h->F));
    CK(dev_alloc(&h->d_hist[0], (size_t)nchan * nsym));
    CK(dev_alloc(&h->d_hist[1], (size_t)nchan * nsym));
    CK(dev_alloc(&h->d_abits, (size_t)nchan * h->abits_stride));
    CK(dev_alloc(&h->d_scratch, (size_t)nchan * h->scratch_stride, false));
    for (int k = 0; k < aisx_corr::NTAGBUF; k++) {
        CK(dev_alloc(&h->d_tags2[k], (size_t)nchan * h->tag_cap));
        CK(dev_alloc(&h->d_tag_count2[k], nchan));
    }
    h->d_tags = h->d_tags2[0];
    h->d_tag_count = h->d_tag_count2[0];
    CK(dev_alloc(&h->d_atan, 257));
    if (hipMemcpy(h->d_wtab, w.data(), sizeof(cf) * h->F, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->d_atan, aisx_atan_table, sizeof(float) * 257, hipMemcpyHostToDevice) != hipSuccess) {
        set_err("aisx_corr_create: table upload failed");
        aisx_corr_destroy(h);
        return AISX_ERR_HIP;
    }
    CK(corr_upload_taps(h));
#undef CK
    *out = h;
    return AISX_OK;
}

extern "C" int aisx_corr_destroy(aisx_corr* h)
{
    if (!h)
        return AISX_OK;
    dev_free(h->d_taps);
    dev_free(h->d_tapspad);
    dev_free(h->d_Hpos);
    dev_free(h->d_wtab);
    dev_free(h->d_hist[0]);
    dev_free(h->d_hist[1]);
    dev_free(h->d_abits);
    dev_free(h->d_scratch);
    for (int k = 0; k < aisx_corr::NTAGBUF; k++) {
        dev_free(h->d_tags2[k]);
        dev_free(h->d_tag_count2[k]);
    }
    dev_free(h->d_atan);
    dev_free(h->d_st_in);
    dev_free(h->d_st_out);
    dev_free(h->d_st_corr);
    for (int k = 0; k < aisx_corr::NEV; k++) {
        if (h->ev0[k])
            (void)hipEventDestroy(h->ev0[k]);
        if (h->ev1[k])
            (void)hipEventDestroy(h->ev1[k]);
    }
    delete h;
    return AISX_OK;
}

extern "C" int aisx_corr_symbols(const aisx_corr* h, aisx_cf32* out, int cap)
{
    if (!h || !out || cap < h->N)
        return AISX_ERR_INVALID;
    memcpy(out, h->symbols.data(), sizeof(cf) * h->N);
    return h->N;
}

extern "C" int aisx_corr_set_symbols(aisx_corr* h, const aisx_cf32* symbols, int nsym)
{
    if (!h || !symbols || nsym < 1)
        return AISX_ERR_INVALID;
    if (nsym > CORR_MAX_TEMPLATE) {
        set_err("aisx_corr_set_symbols: template of %d samples exceeds the %d supported (F = %d build)", nsym,
                CORR_MAX_TEMPLATE, CF4_F);
        return AISX_ERR_INVALID;
    }
    AISX_HIPCHK(hipDeviceSynchronize()); // (the caller holds d_setlock in the reference, :135: no work() in flight)
    int rc;
    if (nsym != h->N) {
        // lib/corr_est_cc_impl.cc:144-158: the FFT filter is rebuilt for the new length, output
        // multiple, history (nsym + 1) and sample delay follow it.  The block's history: the
        // scheduler would hand the nsym items before the next new one; this handle holds the
        // last N of them -- kept right-aligned, older ones (nsym > N) read as zero.
        // Transactional: every new buffer is allocated and filled first; the handle changes only
        // once nothing can fail any more.
        const int Nold = h->N, keep = std::min(Nold, nsym);
        const int F = corr_pick_fft_x(nsym);
        cf *nh[2] = { nullptr, nullptr }, *ntaps = nullptr, *npad = nullptr, *nH = nullptr, *nw = nullptr;
        auto undo = [&](int r) {
            dev_free(nh[0]);
            dev_free(nh[1]);
            dev_free(ntaps);
            dev_free(npad);
            dev_free(nH);
            dev_free(nw);
            return r;
        };
        if ((rc = dev_alloc(&nh[0], (size_t)h->nchan * nsym)) != AISX_OK || (rc = dev_alloc(&nh[1], (size_t)h->nchan * nsym)) != AISX_OK ||
            (rc = dev_alloc(&ntaps, nsym)) != AISX_OK)
            return undo(rc);
        if (F != h->F) {
            if ((rc = dev_alloc(&npad, F)) != AISX_OK || (rc = dev_alloc(&nH, F)) != AISX_OK || (rc = dev_alloc(&nw, F)) != AISX_OK)
                return undo(rc);
            std::vector<cf> w = corr_wtab(F);
            if (hipMemcpy(nw, w.data(), sizeof(cf) * F, hipMemcpyHostToDevice) != hipSuccess) {
                set_err("aisx_corr_set_symbols: twiddle upload failed");
                return undo(AISX_ERR_HIP);
            }
        }
        if (hipMemcpy2D(nh[0] + (nsym - keep), sizeof(cf) * nsym, h->d_hist[h->hist_cur] + (Nold - keep), sizeof(cf) * Nold,
                        sizeof(cf) * keep, h->nchan, hipMemcpyDeviceToDevice) != hipSuccess) {
            set_err("aisx_corr_set_symbols: history copy failed");
            return undo(AISX_ERR_HIP);
        }
        // commit
        dev_free(h->d_hist[0]);
        dev_free(h->d_hist[1]);
        dev_free(h->d_taps);
        h->d_hist[0] = nh[0];
        h->d_hist[1] = nh[1];
        h->d_taps = ntaps;
        h->hist_cur = 0;
        if (F != h->F) {
            dev_free(h->d_tapspad);
            dev_free(h->d_Hpos);
            dev_free(h->d_wtab);
            h->d_tapspad = npad;
            h->d_Hpos = nH;
            h->d_wtab = nw;
            h->F = F;
        }
        h->N = nsym;
        h->L = h->F - nsym;
        h->symbols.resize(nsym);
        // :143-144 set_output_multiple(d_filter->set_taps(d_symbols))
        h->out_multiple = (int)(2 * pow(2.0, ceil(log((double)nsym) / log(2.0)))) - nsym + 1;
    }
    // :136 stored as given (no conjugate / reverse, unlike the constructor :58-63); d_thresh untouched
    memcpy(h->symbols.data(), symbols, sizeof(cf) * nsym);
    h->mark_delay = h->mark_delay >= (unsigned)nsym ? (unsigned)nsym - 1 : h->mark_delay; // :160-161
    // [GR] fft_filter_ccc::set_taps zeroes the filter's tail: the correlation of the next call
    // starts from zeros, the delayed pass-through from the history as before
    h->corr_hist_zero = 1;
    return corr_upload_taps(h);
}

extern "C" int aisx_corr_geometry(const aisx_corr* h, int* nchan, int* max_items)
{
    if (!h)
        return AISX_ERR_INVALID;
    if (nchan)
        *nchan = h->nchan;
    if (max_items)
        *max_items = h->max_items;
    return AISX_OK;
}

extern "C" int aisx_corr_history(const aisx_corr* h) { return h ? h->N + 1 : AISX_ERR_INVALID; }
extern "C" int aisx_corr_output_multiple(const aisx_corr* h) { return h ? h->out_multiple : AISX_ERR_INVALID; }
extern "C" int aisx_corr_max_noutput_items(const aisx_corr*) { return 24 * 1024; }
extern "C" float aisx_corr_threshold(const aisx_corr* h) { return h ? h->thresh : 0.f; }
extern "C" unsigned aisx_corr_mark_delay(const aisx_corr* h) { return h ? h->mark_delay : 0u; }
extern "C" uint64_t aisx_corr_nitems_written(const aisx_corr* h) { return h ? h->written : 0; }

extern "C" int aisx_corr_reset(aisx_corr* h)
{
    if (!h)
        return AISX_ERR_INVALID;
    AISX_HIPCHK(hipMemset(h->d_hist[0], 0, sizeof(cf) * (size_t)h->nchan * h->N));
    AISX_HIPCHK(hipMemset(h->d_hist[1], 0, sizeof(cf) * (size_t)h->nchan * h->N));
    AISX_HIPCHK(hipDeviceSynchronize()); // (null-stream fill vs. the caller's non-blocking streams)
    h->hist_cur = 0;
    h->written = 0;
    return AISX_OK;
}

extern "C" int aisx_corr_process(aisx_corr* h, const aisx_cf32* d_in, long in_stride, aisx_cf32* d_out,
                                 long out_stride, aisx_cf32* d_corr, long corr_stride, int n, void* stream)
{
    if (!h || !d_in || !d_out || n < 1 || n > h->max_items || in_stride < n || out_stride < n ||
        (d_corr && corr_stride < n)) {
        set_err("aisx_corr_process: bad argument (n=%d, max_items=%d)", n, h ? h->max_items : -1);
        return AISX_ERR_INVALID;
    }
    {
        // out is the input delayed by N and is written while neighbouring tiles are still being
        // read (the resolver reads d_in once more after that): the three buffers must be distinct
        auto span = [&](const void* b, long stride) {
            const char* lo = (const char*)b;
            return std::make_pair(lo, lo + sizeof(cf) * ((size_t)(h->nchan - 1) * (size_t)stride + (size_t)n));
        };
        auto overlap = [](std::pair<const char*, const char*> a, std::pair<const char*, const char*> b) {
            return a.first < b.second && b.first < a.second;
        };
        const auto ri = span(d_in, in_stride), ro = span(d_out, out_stride);
        if (overlap(ri, ro) || (d_corr && (overlap(ri, span(d_corr, corr_stride)) || overlap(ro, span(d_corr, corr_stride))))) {
            set_err("aisx_corr_process: d_in, d_out and d_corr must not overlap (no in-place operation)");
            return AISX_ERR_INVALID;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const bool dma = h->dma; // the builds with the window prefetched by LDS-DMA (k_corr4d.h, k_corr2d.h)
    int nseg, tps;
    corr_grid(h->nchan, n, h->L, h->F, &nseg, &tps, dma ? (h->F == CF4_F ? 2 : 4) : 0);
    static const int force_nseg = [] { // (experiments: segments per channel, AISX_CORR_NSEG)
        const char* e = exp_env("AISX_CORR_NSEG");
        return e ? atoi(e) : 0;
    }();
    if (force_nseg > 0) {
        const int ntiles = (n + h->L - 1) / h->L;
        tps = (ntiles + force_nseg - 1) / force_nseg;
        nseg = (ntiles + tps - 1) / tps;
    }

    AISX_HIPCHK(hipMemsetAsync(h->d_abits, 0, sizeof(unsigned long long) * (size_t)h->nchan * h->abits_stride, st));
    CorrParams p;
    p.in = (const cf*)d_in;
    p.in_stride = in_stride;
    p.out = (cf*)d_out;
    p.out_stride = out_stride;
    p.corr = d_corr ? (cf*)d_corr : h->d_scratch;
    p.corr_stride = d_corr ? corr_stride : h->scratch_stride;
    p.dense_corr = d_corr ? 1 : 0;
    p.hist_in = h->d_hist[h->hist_cur];
    p.hist_out = h->d_hist[h->hist_cur ^ 1];
    p.Hpos = h->d_Hpos;
    p.wtab = h->d_wtab;
    p.abits = h->d_abits;
    p.abits_stride = h->abits_stride;
    p.n = n;
    p.N = h->N;
    p.L = h->L;
    p.nseg = nseg;
    p.tiles_per_seg = tps;
    p.thresh = h->thresh;
    p.corr_hist_zero = h->corr_hist_zero;
    const int evi = (int)(h->ncalls_prof % aisx_corr::NEV);
    if (h->prof)
        AISX_HIPCHK(hipEventRecord(h->ev0[evi], st));
    int rc_launch = AISX_OK;
    auto launch_big_lds = [&](void (*kern)(CorrParams), int threads, int lds_bytes) -> int {
        if (h->dma_attr_set != (const void*)kern || lds_bytes > h->dma_attr_bytes) { // (per handle: handles may live on different devices)
            AISX_HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
            h->dma_attr_set = (const void*)kern;
            h->dma_attr_bytes = lds_bytes;
        }
        hipLaunchKernelGGL(kern, dim3(nseg, h->nchan), dim3(threads), lds_bytes, st, p);
        return AISX_OK;
    };
#ifdef AISX_EXPERIMENTS
    // (LDS a workgroup claims beyond what it uses decides how many of them fit beside the timing recovery's 92 160 bytes on a CU)
    static const int lds_pad_env = exp_env("AISX_CORR_LDS_PAD") ? atoi(exp_env("AISX_CORR_LDS_PAD")) : 0;
    const int lds_pad = lds_pad_env + h->lds_claim;
    if (h->F == CF_F && !dma)
        hipLaunchKernelGGL(k_corr_main, dim3(nseg, h->nchan), dim3(CF_T), CF_LDS_BYTES, st, p);
    else if (h->F == CF4_F && !dma)
        hipLaunchKernelGGL(k_corr4_main, dim3(nseg, h->nchan), dim3(CF4_T), CF4_LDS_BYTES, st, p);
    else if (h->F == CF4_F && h->wide == 0)
        rc_launch = launch_big_lds(corr4d_pick(h->N), CF4_T, CD_LDS_BYTES + lds_pad);
    else if (h->F == CF4_F && h->wide == 1)
        rc_launch = launch_big_lds(corr4e_pick(h->N), CE_T, CE_LDS_BYTES + lds_pad);
    else
#else
    const int lds_pad = h->lds_claim;
#endif
    if (h->F == CF_F)
        hipLaunchKernelGGL(corr2d_pick(h->N), dim3(nseg, h->nchan), dim3(CF_T), C2_LDS_BYTES, st, p);
    else
        rc_launch = launch_big_lds(corr4f_pick(h->N), CE_T, cfz_lds_bytes(h->N) + lds_pad);
    if (rc_launch != AISX_OK)
        return rc_launch;
    AISX_HIPCHK(hipGetLastError());
    if (h->prof) {
        AISX_HIPCHK(hipEventRecord(h->ev1[evi], st));
        h->ncalls_prof++;
    }

    ResolveParams r;
    r.abits = h->d_abits;
    r.abits_stride = h->abits_stride;
    r.L = h->L;
    r.corr = p.corr;
    r.corr_stride = p.corr_stride;
    r.dense_corr = p.dense_corr;
    r.in = p.in;
    r.in_stride = in_stride;
    r.hist_in = p.hist_in;
    r.corr_hist_zero = h->corr_hist_zero;
    r.taps = h->d_taps;
    r.n = n;
    r.N = h->N;
    r.isps = h->isps;
    r.mark_delay = h->mark_delay;
    r.written = h->written;
    r.emit_port1 = d_corr ? 1 : 0;
    h->tag_cur = (h->tag_cur + 1) % aisx_corr::NTAGBUF;
    h->d_tags = h->d_tags2[h->tag_cur];
    h->d_tag_count = h->d_tag_count2[h->tag_cur];
    r.tags = h->d_tags;
    r.tag_cap = h->tag_cap;
    r.tag_count = h->d_tag_count;
    r.atan_tab = h->d_atan;
    {
        const int nwv = rsv_waves_for(h->nchan);
        hipLaunchKernelGGL(k_corr_resolve, dim3(h->nchan), dim3(64 * nwv), rsv_lds_bytes(nwv), st, r);
    }
    AISX_HIPCHK(hipGetLastError());
    h->hist_cur ^= 1;
    h->corr_hist_zero = 0;
    h->written += (uint64_t)n;
    h->last_emit_port1 = r.emit_port1;
    return AISX_OK;
}

extern "C" int aisx_corr_set_profiling(aisx_corr* h, int on)
{
    if (!h)
        return AISX_ERR_INVALID;
    if (on && !h->ev0[0]) {
        for (int k = 0; k < aisx_corr::NEV; k++) {
            AISX_HIPCHK(hipEventCreate(&h->ev0[k]));
            AISX_HIPCHK(hipEventCreate(&h->ev1[k]));
        }
    }
    h->prof = on ? 1 : 0;
    h->ncalls_prof = 0;
    return AISX_OK;
}

extern "C" int aisx_corr_last_kernel_ms(aisx_corr* h, float* ms)
{
    if (!h || !ms || !h->ev0[0] || h->ncalls_prof < 1)
        return AISX_ERR_INVALID;
    const int evi = (int)((h->ncalls_prof - 1) % aisx_corr::NEV);
    AISX_HIPCHK(hipEventSynchronize(h->ev1[evi]));
    AISX_HIPCHK(hipEventElapsedTime(ms, h->ev0[evi], h->ev1[evi]));
    return AISX_OK;
}

extern "C" int aisx_corr_kernel_ms_history(aisx_corr* h, float* ms, int cap, int* n)
{
    if (!h || !ms || !n || !h->ev0[0])
        return AISX_ERR_INVALID;
    const long have = h->ncalls_prof < aisx_corr::NEV ? h->ncalls_prof : aisx_corr::NEV;
    int w = 0;
    for (long k = h->ncalls_prof - have; k < h->ncalls_prof && w < cap; k++) {
        const int evi = (int)(k % aisx_corr::NEV);
        AISX_HIPCHK(hipEventSynchronize(h->ev1[evi]));
        AISX_HIPCHK(hipEventElapsedTime(&ms[w], h->ev0[evi], h->ev1[evi]));
        w++;
    }
    *n = w;
    return AISX_OK;
}

extern "C" int aisx_corr_set_lds_claim(aisx_corr* h, int bytes)
{
    if (!h || bytes < 0 || bytes > 96 * 1024) {
        set_err("aisx_corr_set_lds_claim: 0 .. 98304 bytes");
        return AISX_ERR_INVALID;
    }
    h->lds_claim = bytes;
    return AISX_OK;
}

extern "C" int aisx_corr_tags_device(const aisx_corr* h, const aisx_tag** d_tags, const int** d_counts, int* cap)
{
    if (!h)
        return AISX_ERR_INVALID;
    if (d_tags)
        *d_tags = (const aisx_tag*)h->d_tags;
    if (d_counts)
        *d_counts = h->d_tag_count;
    if (cap)
        *cap = h->tag_cap;
    return AISX_OK;
}

extern "C" int aisx_corr_read_tags_back(aisx_corr* h, int back, aisx_tag* host_tags, int host_cap, int* ntags, void* stream)
{
    if (!h || !ntags || back < 0 || back >= aisx_corr::NTAGBUF)
        return AISX_ERR_INVALID;
    const int bi = (h->tag_cur + aisx_corr::NTAGBUF - back) % aisx_corr::NTAGBUF;
    const tag_rec* d_tags = h->d_tags2[bi];
    const int* d_tag_count = h->d_tag_count2[bi];
    hipStream_t st = (hipStream_t)stream;
    std::vector<int> counts(h->nchan);
    AISX_HIPCHK(hipMemcpyAsync(counts.data(), d_tag_count, sizeof(int) * h->nchan, hipMemcpyDeviceToHost, st));
    AISX_HIPCHK(hipStreamSynchronize(st));
    int total = 0, rc = AISX_OK;
    long maxc = 0;
    for (int c = 0; c < h->nchan; c++) {
        if (counts[c] > h->tag_cap) {
            rc = AISX_ERR_OVERFLOW;
            counts[c] = h->tag_cap;
        }
        maxc = std::max<long>(maxc, counts[c]);
    }
    if (maxc > 0 && host_tags) {
        // one strided copy of the used prefix of every channel's segment
        std::vector<tag_rec> tmp((size_t)h->nchan * maxc);
        AISX_HIPCHK(hipMemcpy2DAsync(tmp.data(), sizeof(tag_rec) * maxc, d_tags, sizeof(tag_rec) * h->tag_cap,
                                     sizeof(tag_rec) * maxc, h->nchan, hipMemcpyDeviceToHost, st));
        AISX_HIPCHK(hipStreamSynchronize(st));
        for (int c = 0; c < h->nchan; c++)
            for (int k = 0; k < counts[c]; k++) {
                if (total < host_cap)
                    memcpy(&host_tags[total], &tmp[(size_t)c * maxc + k], sizeof(tag_rec));
                else
                    rc = AISX_ERR_OVERFLOW;
                total++;
            }
    } else {
        for (int c = 0; c < h->nchan; c++)
            total += counts[c];
        if (total > 0 && !host_tags)
            rc = AISX_ERR_OVERFLOW;
    }
    *ntags = std::min(total, host_tags ? host_cap : 0);
    if (rc == AISX_ERR_OVERFLOW)
        set_err("aisx_corr_read_tags: tag buffer overflow (%d tags)", total);
    return rc;
}

extern "C" int aisx_corr_read_tags(aisx_corr* h, aisx_tag* host_tags, int host_cap, int* ntags, void* stream)
{
    return aisx_corr_read_tags_back(h, 0, host_tags, host_cap, ntags, stream);
}

extern "C" int aisx_corr_work_host(aisx_corr* h, const aisx_cf32* in, aisx_cf32* out, aisx_cf32* corr,
                                   int noutput_items, uint64_t nitems_written, aisx_tag* tags, int tag_cap,
                                   int* ntags)
{
    if (!h || !in || !out || noutput_items < 1 || !ntags)
        return AISX_ERR_INVALID;
    if (h->nchan != 1) {
        set_err("aisx_corr_work_host: handle has %d channels, the GNU Radio path needs 1", h->nchan);
        return AISX_ERR_INVALID;
    }
    const int n = noutput_items;
    if (n > h->st_cap) {
        dev_free(h->d_st_in);
        dev_free(h->d_st_out);
        dev_free(h->d_st_corr);
        int rc = dev_alloc(&h->d_st_in, n, false);
        if (rc == AISX_OK)
            rc = dev_alloc(&h->d_st_out, n, false);
        if (rc == AISX_OK)
            rc = dev_alloc(&h->d_st_corr, n, false);
        if (rc != AISX_OK)
            return rc;
        h->st_cap = n;
    }
    // in[0 .. N) is the block's history, in[N .. N+n) the new items (:180-188)
    AISX_HIPCHK(hipMemcpy(h->d_hist[h->hist_cur], in, sizeof(cf) * h->N, hipMemcpyHostToDevice));
    AISX_HIPCHK(hipMemcpy(h->d_st_in, in + h->N, sizeof(cf) * n, hipMemcpyHostToDevice));
    h->written = nitems_written;
    int rc = aisx_corr_process(h, (aisx_cf32*)h->d_st_in, n, (aisx_cf32*)h->d_st_out, n,
                               corr ? (aisx_cf32*)h->d_st_corr : nullptr, n, n, nullptr);
    if (rc != AISX_OK)
        return rc;
    AISX_HIPCHK(hipMemcpy(out, h->d_st_out, sizeof(cf) * n, hipMemcpyDeviceToHost));
    if (corr)
        AISX_HIPCHK(hipMemcpy(corr, h->d_st_corr, sizeof(cf) * n, hipMemcpyDeviceToHost));
    return aisx_corr_read_tags(h, tags, tag_cap, ntags, nullptr);
}

