// emul.cpp -- CPU model of the execution context the HIP kernel bodies run
// under (one OS thread per lane, std::barrier for __syncthreads / wave ops).
// TEST INFRASTRUCTURE: it instantiates the *same* kernel-body templates the
// product compiles for gfx950 (gr-ais_amd/csrc/k_*.h) so their index arithmetic
// can be checked against the oracle on a machine without a GPU.  It is not part
// of libaisx.so and nothing in the product can reach it.
#include <barrier>
#include <memory>
#include <cstring>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>

#include "../../gr-ais_amd/csrc/aisx_common.h"
#include "../../gr-ais_amd/csrc/aisx_tables.h"
#include "../../gr-ais_amd/csrc/k_corr.h"
#include "../../gr-ais_amd/csrc/k_msk.h"
#include "../../gr-ais_amd/csrc/k_mskp.h"
#ifdef MSK_EMU_STATS
namespace aisx { long msk_stats[8]; }
#endif
#include "../../gr-ais_amd/csrc/aisx_plan.h"
#include "../../gr-ais_amd/csrc/k_pfb.h"
#if __has_include("../../gr-ais_amd/csrc/k_agc.h")
#include "../../gr-ais_amd/csrc/k_agc.h"
#include "../../gr-ais_amd/csrc/k_agcw.h"
#define HAVE_AGC 1
#endif
#if __has_include("../../gr-ais_amd/csrc/k_freqsync.h")
#include "../../gr-ais_amd/csrc/k_freqsync.h"
#define HAVE_FREQSYNC 1
#endif

using namespace aisx;

struct EmuShared {
    std::barrier<> bar;                                // workgroup barrier
    std::vector<std::unique_ptr<std::barrier<>>> wbar; // one per wave: ballots / lane exchanges
    int nthreads;
    std::vector<char> lds;
    unsigned long long x64[2048] = {};
    explicit EmuShared(int nt, size_t ldsbytes) : bar(nt), nthreads(nt), lds(ldsbytes + 64)
    {
        for (int w = 0; w * 64 < nt; w++)
            wbar.emplace_back(new std::barrier<>(std::min(64, nt - w * 64)));
    }
};

struct EmuCtx {
    static constexpr bool wave_lds_coherent = false; // (lanes are free-running threads: only a barrier orders their LDS accesses)
    EmuShared* sh;
    int tid_, bx_, by_;
    int tid() const { return tid_; }
    int nthreads() const { return sh->nthreads; }
    int bx() const { return bx_; }
    int by() const { return by_; }
    char* lds() const { return sh->lds.data(); }
    void sync() const { sh->bar.arrive_and_wait(); }
    void wsync() const { sh->wbar[tid_ >> 6]->arrive_and_wait(); }
    void wave_sync() const { wsync(); }
    void wave_lds_sync() const { wsync(); } // (only the wave's own lanes meet: a cross-wave exchange would race here)
    // a lane that leaves the kernel for good: it stops counting in the barriers
    void retire() const
    {
        sh->wbar[tid_ >> 6]->arrive_and_drop();
        sh->bar.arrive_and_drop();
    }
    int wave_base() const { return tid_ & ~63; }
    // wave collectives: one barrier per call; the exchange slots are double-buffered
    // (bank = parity of this lane's call count; all lanes of a block call in lock-step)
    mutable unsigned ncall = 0;
    unsigned long long* bank() const { return sh->x64 + ((ncall++ & 1u) ? 1024 : 0); }
    unsigned long long ballot(bool p) const
    {
        unsigned long long* b = bank();
        b[tid_] = p ? 1ull : 0ull;
        wsync();
        unsigned long long m = 0;
        for (int l = 0; l < 64 && wave_base() + l < sh->nthreads; l++)
            m |= b[wave_base() + l] << l;
        return m;
    }
    float fract(float x) const { return x - floorf(x); }
    // complex primitives of the FFT butterflies: the arithmetic of the packed instructions
    cf cadd(cf a, cf b) const { return mk(a.re + b.re, a.im + b.im); }
    cf csub(cf a, cf b) const { return mk(a.re - b.re, a.im - b.im); }
    cf add_mj(cf a, cf b) const { return mk(a.re + b.im, a.im - b.re); }
    cf add_pj(cf a, cf b) const { return mk(a.re - b.im, a.im + b.re); }
    cf cmul(cf a, cf b) const { return cmul_fma(a, b); }
    cf cmul_conj(cf a, cf b) const { return cmul_conj_fma(a, b); }
    template <int KSEL, int CS, bool CNEG, int SS, bool SNEG>
    cf cmul_sel(cf a) const
    {
        const float K[2][2] = { { 0.92387953251128673848f, 0.38268343236508978178f },
                                { 0.70710678118654752440f, 0.70710678118654752440f } };
        const float c = CNEG ? -K[KSEL][CS] : K[KSEL][CS], s = SNEG ? -K[KSEL][SS] : K[KSEL][SS];
        return mk(fmaf(-a.im, s, a.re * c), fmaf(a.re, s, a.im * c));
    }
    void store16(cf* base, unsigned off, cf a, cf b) const
    {
        cf* d = (cf*)((char*)base + off);
        d[0] = a;
        d[1] = b;
    }
    float sel_f32(unsigned long long m, float if_set, float if_clear) const { return ((m >> (tid_ & 63)) & 1ull) ? if_set : if_clear; }
    void pin(float&) const {}
    void pin(int&) const {}
    void pin_mask(unsigned long long&) const {}
    bool inv_ballot(unsigned long long m) const { return (m >> (tid_ & 63)) & 1ull; }
    template <class T>
    T xchg(T v, int src_lane) const
    {
        static_assert(sizeof(T) <= 8, "");
        unsigned long long raw = 0;
        memcpy(&raw, &v, sizeof(T));
        unsigned long long* b = bank();
        b[tid_] = raw;
        wsync();
        int s = wave_base() + (src_lane & 63);
        if (s >= sh->nthreads)
            s = tid_;
        unsigned long long r = b[s];
        T out;
        memcpy(&out, &r, sizeof(T));
        return out;
    }
    // lo = v of lane (i & ~W), hi = v of lane (i | W)
    template <int W>
    void pair_rows(cf v, cf& lo, cf& hi) const
    {
        unsigned long long raw = 0;
        memcpy(&raw, &v, sizeof(cf));
        unsigned long long* b = bank();
        b[tid_] = raw;
        wsync();
        const int ln = tid_ & 63;
        memcpy(&lo, &b[wave_base() + (ln & ~W)], sizeof(cf));
        memcpy(&hi, &b[wave_base() + (ln | W)], sizeof(cf));
    }
    // wave scans of non-negative integers (DevCtx: DPP), 0 the identity
    int wave_excl_prefix_max_nn(int v) const
    {
        unsigned long long* b = bank();
        b[tid_] = (unsigned long long)(unsigned)v;
        wsync();
        int r = 0;
        for (int i = 0; i < (tid_ & 63); i++)
            r = std::max(r, (int)(unsigned)b[wave_base() + i]);
        return r;
    }
    int wave_excl_suffix_max_nn(int v, int& all) const
    {
        unsigned long long* b = bank();
        b[tid_] = (unsigned long long)(unsigned)v;
        wsync();
        int r = 0;
        for (int i = (tid_ & 63) + 1; i < 64 && wave_base() + i < sh->nthreads; i++)
            r = std::max(r, (int)(unsigned)b[wave_base() + i]);
        all = r;
        for (int i = 0; i <= (tid_ & 63); i++)
            all = std::max(all, (int)(unsigned)b[wave_base() + i]);
        return r;
    }
    float rcp_approx(float x) const { return 1.0f / x; } // (the lane model's is correctly rounded; the device's is within 1 ulp)
    float sqrt_approx(float x) const { return sqrtf(x); }
    cf lds_cf(const float* tab, unsigned idx) const { return ld8(reinterpret_cast<const cf*>(tab) + idx); }
    unsigned lane_prev_u32(unsigned v) const { const int l = tid_ & 63; const unsigned r = xchg(v, l > 0 ? l - 1 : l); return l > 0 ? r : 0u; }
    unsigned lane_next_u32(unsigned v) const { const int l = tid_ & 63; const unsigned r = xchg(v, l < 63 ? l + 1 : l); return l < 63 ? r : 0u; }
    // (aisx_devctx.h: xpose8_lane_hi) x[j] of lane (h, e) <- x[h] of lane (j, e)
    void xpose8_lane_hi(cf (&x)[8]) const
    {
        const int l = tid_ & 63, h = (l >> 3) & 7, e = l & 7;
        cf y[8];
        for (int j = 0; j < 8; j++) {
            // every lane offers, for this round, the register the asking lanes (j' = h of the source) want: lane (j, e) is
            // asked by lane (h, e) for its x[h]; in round j the source lane of lane (h, e) is (j, e), and that lane must
            // publish x[h of the asker] -- askers of one source all differ in h, so publish one register per round r and
            // let lane (h, e) pick in round r = h from source (j, e): run the 8 x 8 rounds
            y[j] = mk(0.f, 0.f);
        }
        for (int r = 0; r < 8; r++)     // register published this round
            for (int j = 0; j < 8; j++) { // source lane's high bits
                const float re = xchg(x[r].re, (j << 3) | e), im = xchg(x[r].im, (j << 3) | e);
                if (r == h)
                    y[j] = mk(re, im);
            }
        for (int j = 0; j < 8; j++)
            x[j] = y[j];
    }
    unsigned long long shfl_u64(unsigned long long v, int src) const { return xchg(v, src); }
    float shfl_f32(float v, int src) const { return xchg(v, src); }
    int shfl_i32(int v, int src) const { return xchg(v, src); }
    float shfl_up_f32(float v, int d) const { int l = tid_ & 63; return xchg(v, l - d >= 0 ? l - d : l); }
    float shfl_down_f32(float v, int d) const { int l = tid_ & 63; return xchg(v, l + d < 64 ? l + d : l); }
    float shfl_xor_f32(float v, int m) const { return xchg(v, (tid_ & 63) ^ m); }
    int shfl_xor_i32(int v, int m) const { return xchg(v, (tid_ & 63) ^ m); }
    double shfl_xor_f64(double v, int m) const { return xchg(v, (tid_ & 63) ^ m); }
    int readlane_i32(int v, int lane) const { return xchg(v, lane); }
    int ctz64(unsigned long long v) const { return __builtin_ctzll(v); }
    void atomic_or64(unsigned long long* p, unsigned long long v) const { __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
    int atomic_add_i32(int* p, int v) const { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
    // raw buffers and LDS-DMA (model of DevCtx's: the copy is done on the spot)
    struct Buf {
        const void* base;
        unsigned nbytes;
    };
    int wave_id() const { return tid_ >> 6; }
    Buf make_buf(const void* base, unsigned nbytes) const { return Buf{ base, nbytes }; }
    unsigned lds_addr(const void* p) const { return (unsigned)((const char*)p - sh->lds.data()); }
    void dma16(const Buf& b, unsigned byte_off, unsigned lds_dst) const
    {
        char* d = sh->lds.data() + lds_dst + 16 * (tid_ & 63);
        for (int k = 0; k < 4; k++) { // dword granularity of the range check
            const unsigned o = byte_off + 4u * k;
            // (an access that starts out of range is taken as out of range altogether, also where
            // its 32-bit offset would wrap back into the buffer: the pessimistic reading)
            if (byte_off < b.nbytes && o < b.nbytes && o + 4u <= b.nbytes)
                memcpy(d + 4 * k, (const char*)b.base + o, 4);
            else
                memset(d + 4 * k, 0, 4);
        }
    }
    void wait_dma() const {}
    template <int N>
    void wait_vm() const {}
    void lds_barrier() const { sync(); }
    cf buf_load64(const Buf& b, unsigned voff, unsigned soff) const
    {
        cf v = mk(0.f, 0.f);
        if (voff < b.nbytes && voff + 8u <= b.nbytes)
            memcpy(&v, (const char*)b.base + voff + soff, 8);
        return v;
    }
    void buf_store64(const Buf& b, unsigned voff, unsigned soff, cf v) const
    {
        if (voff < b.nbytes && voff + 8u <= b.nbytes)
            memcpy((char*)b.base + voff + soff, &v, 8);
    }
};

template <class Body>
static void run_grid(int gx, int gy, int nthreads, size_t ldsbytes, Body body)
{
    for (int by = 0; by < gy; by++)
        for (int bx = 0; bx < gx; bx++) {
            EmuShared sh(nthreads, ldsbytes);
            std::vector<std::thread> th;
            th.reserve(nthreads);
            for (int t = 0; t < nthreads; t++)
                th.emplace_back([&, t]() {
                    EmuCtx cx{ &sh, t, bx, by };
                    body(cx);
                });
            for (auto& x : th)
                x.join();
        }
}

// lanes that never synchronise can simply run one after another
template <class Body>
static void run_independent(int gx, int nthreads, Body body)
{
    EmuShared sh(1, 0);
    sh.nthreads = nthreads;
    for (int bx = 0; bx < gx; bx++)
        for (int t = 0; t < nthreads; t++) {
            EmuCtx cx{ &sh, t, bx, 0 };
            body(cx);
        }
}

extern "C" {

int emu_cf_sizes(int* F, int* T) { *F = CF_F; *T = CF_T; return CF_LDS_BYTES; }
int emu_corr_max_template() { return CORR_MAX_TEMPLATE; }

// which F = 4096 build runs: 0 = k_corr4k.h, 1 = k_corr4d.h (the length-specialised build where
// there is one, as the product picks), 2 = k_corr4d.h's run-time-length build, 3 / 4 = the same two
// of k_corr4e.h (512 threads x 8 points; the template spectrum is in ITS position order: the mode
// is read when a handle is created), 5 / 6 = the same two of k_corr4f.h (the next window in registers)
static int g_corr_dma = 1;
void emu_corr_set_dma(int mode) { g_corr_dma = mode; }

void emu_corr_inith(const cf* taps_scaled, const cf* wtab, cf* Hpos, int F)
{
    CorrInitParams p{ taps_scaled, wtab, Hpos };
    if (F == CF_F)
        run_grid(1, 1, CF_T, CF_LDS_BYTES, [&](EmuCtx& cx) { corr_inith_body(cx, p); });
    else if (g_corr_dma >= 3)
        run_grid(1, 1, CE_T, CE_LDS_BYTES, [&](EmuCtx& cx) { corr4e_inith_body(cx, p); });
    else
        run_grid(1, 1, CF4_T, CF4_LDS_BYTES, [&](EmuCtx& cx) { corr4_inith_body(cx, p); });
}

void emu_corr_main(const CorrParams* p, int nchan, int F)
{
    if (F == CF_F && g_corr_dma == 0)
        run_grid(p->nseg, nchan, CF_T, CF_LDS_BYTES, [&](EmuCtx& cx) { corr_main_body(cx, *p); });
    else if (F == CF_F && g_corr_dma == 1 && p->N == 112)
        run_grid(p->nseg, nchan, CF_T, C2_LDS_BYTES, [&](EmuCtx& cx) { corr2d_main_body<EmuCtx, 112>(cx, *p); });
    else if (F == CF_F && g_corr_dma == 1 && p->N == 140)
        run_grid(p->nseg, nchan, CF_T, C2_LDS_BYTES, [&](EmuCtx& cx) { corr2d_main_body<EmuCtx, 140>(cx, *p); });
    else if (F == CF_F)
        run_grid(p->nseg, nchan, CF_T, C2_LDS_BYTES, [&](EmuCtx& cx) { corr2d_main_body<EmuCtx, 0>(cx, *p); });
    else if (g_corr_dma == 0)
        run_grid(p->nseg, nchan, CF4_T, CF4_LDS_BYTES, [&](EmuCtx& cx) { corr4_main_body(cx, *p); });
    else if (g_corr_dma == 3 && p->N == 896)
        run_grid(p->nseg, nchan, CE_T, CE_LDS_BYTES, [&](EmuCtx& cx) { corr4e_main_body<EmuCtx, 896>(cx, *p); });
    else if (g_corr_dma == 3 && p->N == 1120)
        run_grid(p->nseg, nchan, CE_T, CE_LDS_BYTES, [&](EmuCtx& cx) { corr4e_main_body<EmuCtx, 1120>(cx, *p); });
    else if (g_corr_dma == 3 || g_corr_dma == 4)
        run_grid(p->nseg, nchan, CE_T, CE_LDS_BYTES, [&](EmuCtx& cx) { corr4e_main_body<EmuCtx, 0>(cx, *p); });
    else if (g_corr_dma == 5 && p->N == 896)
        run_grid(p->nseg, nchan, CE_T, cfz_lds_bytes(p->N), [&](EmuCtx& cx) { corr4f_main_body<EmuCtx, 896>(cx, *p); });
    else if (g_corr_dma == 5 && p->N == 1120)
        run_grid(p->nseg, nchan, CE_T, cfz_lds_bytes(p->N), [&](EmuCtx& cx) { corr4f_main_body<EmuCtx, 1120>(cx, *p); });
    else if (g_corr_dma >= 5)
        run_grid(p->nseg, nchan, CE_T, cfz_lds_bytes(p->N), [&](EmuCtx& cx) { corr4f_main_body<EmuCtx, 0>(cx, *p); });
    else if (g_corr_dma == 1 && p->N == 896)
        run_grid(p->nseg, nchan, CF4_T, CD_LDS_BYTES, [&](EmuCtx& cx) { corr4d_main_body<EmuCtx, 896>(cx, *p); });
    else if (g_corr_dma == 1 && p->N == 1120)
        run_grid(p->nseg, nchan, CF4_T, CD_LDS_BYTES, [&](EmuCtx& cx) { corr4d_main_body<EmuCtx, 1120>(cx, *p); });
    else
        run_grid(p->nseg, nchan, CF4_T, CD_LDS_BYTES, [&](EmuCtx& cx) { corr4d_main_body<EmuCtx, 0>(cx, *p); });
}

void emu_corr_resolve(const ResolveParams* p, int nchan)
{
    // (the lane model takes sixteen waves whatever the channel count, or what EMU_RSV_WAVES says: the region logic is what it is there to test)
    const char* e = getenv("EMU_RSV_WAVES");
    const int nwv = e ? atoi(e) : RSV_WAVES;
    run_grid(nchan, 1, 64 * nwv, rsv_lds_bytes(nwv), [&](EmuCtx& cx) { corr_resolve_body(cx, *p); });
}

#ifdef MSK_EMU_STATS
long* emu_msk_stats() { return aisx::msk_stats; }
#endif
void emu_bittail(const BitTailParams* p, int max_out)
{
    const int nseg = std::max(1, (max_out + BT_SEG - 1) / BT_SEG);
    run_grid(nseg, p->nchan, BT_T, 260 * 4, [&](EmuCtx& cx) { bittail_body(cx, *p); });
}

// the bit tail on its own: one channel, `n` symbols, state (prev symbol, prev sliced bit) in / out
void emu_bittail_run(const cf* syms, int n, cf* prev_sym, unsigned char* prev_bit, unsigned char* bits)
{
    BitTailParams b;
    cf so = *prev_sym;
    unsigned char bo = *prev_bit;
    b.nchan = 1;
    b.syms = syms; b.sym_stride = n;
    b.produced = &n;
    b.bits = bits; b.bit_stride = n;
    b.prev_sym_in = prev_sym; b.prev_bit_in = prev_bit;
    b.prev_sym_out = &so; b.prev_bit_out = &bo;
    b.atan_tab = aisx_atan_table;
    emu_bittail(&b, n);
    *prev_sym = so;
    *prev_bit = bo;
}

void emu_msk(const MskParams* p)
{
    const bool aux = p->err || p->mu_out;
    auto go = [&](auto lpw_tag) {
        constexpr int L = decltype(lpw_tag)::value;
        run_grid((p->nchan + msk_wg_channels(L) - 1) / msk_wg_channels(L), 1, 64 * msk_waves(L), p->lds_tab_off + MSK_LDS_MMSE, [&](EmuCtx& cx) {
            if (p->ff)
                msk_body<EmuCtx, false, false, L, true>(cx, *p);
            else if (p->osps == 2)
                aux ? msk_body<EmuCtx, true, true, L>(cx, *p) : msk_body<EmuCtx, false, true, L>(cx, *p);
            else
                aux ? msk_body<EmuCtx, true, false, L>(cx, *p) : msk_body<EmuCtx, false, false, L>(cx, *p);
        });
    };
    if (p->lpw == 16)
        go(std::integral_constant<int, 16>{});
    else if (p->lpw == 32)
        go(std::integral_constant<int, 32>{});
    else if (p->lpw == 8)
        go(std::integral_constant<int, 8>{});
    else if (p->lpw == 4)
        go(std::integral_constant<int, 4>{});
    else
        go(std::integral_constant<int, 64>{});
}

// ---- corr_est_cc handle mirroring aisx_corr_* (host orchestration of aisx_lib.hip) ----
struct EmuCorr {
    CorrSetup cs;
    int N, nchan, L, F;
    std::vector<cf> wtab, Hpos, hist[2], scratch;
    std::vector<unsigned long long> abits;
    int cur = 0;
    unsigned long long written = 0;
};

void* emu_corr_create(const cf* symbols, int nsym, float sps, unsigned mark_delay, float threshold, int nchan)
{
    EmuCorr* h = new EmuCorr();
    h->cs = corr_setup(symbols, nsym, sps, mark_delay, threshold);
    h->N = nsym;
    h->nchan = nchan;
    h->F = corr_pick_fft(nsym);
    h->L = h->F - nsym;
    h->wtab = corr_wtab(h->F);
    h->Hpos.resize(h->F);
    std::vector<cf> pad = corr_padded_taps(h->cs.symbols, h->F);
    emu_corr_inith(pad.data(), h->wtab.data(), h->Hpos.data(), h->F);
    h->hist[0].assign((size_t)nchan * nsym, mk(0, 0));
    h->hist[1].assign((size_t)nchan * nsym, mk(0, 0));
    return h;
}
void emu_corr_destroy(void* hv) { delete (EmuCorr*)hv; }
float emu_corr_threshold(void* hv) { return ((EmuCorr*)hv)->cs.thresh; }
int emu_corr_output_multiple(void* hv) { return ((EmuCorr*)hv)->cs.out_multiple; }
void emu_corr_symbols(void* hv, cf* out) { EmuCorr* h = (EmuCorr*)hv; memcpy(out, h->cs.symbols.data(), sizeof(cf) * h->N); }

int emu_corr_process(void* hv, const cf* in, long in_stride, cf* out, long out_stride, cf* corr, long corr_stride,
                     int n, tag_rec* tags, int tag_cap, int* tag_count, int force_nseg)
{
    EmuCorr* h = (EmuCorr*)hv;
    int nseg, tps;
    corr_grid(h->nchan, n, h->L, h->F, &nseg, &tps, (h->F == CF4_F && g_corr_dma) ? 2 : 0);
    if (force_nseg > 0) {
        const int ntiles = (n + h->L - 1) / h->L;
        tps = (ntiles + force_nseg - 1) / force_nseg;
        nseg = (ntiles + tps - 1) / tps;
    }
    const long astride = (n + 63) / 64 + 1;
    h->abits.assign((size_t)h->nchan * astride, 0ull);
    h->scratch.assign((size_t)h->nchan * n, mk(-7777.f, -7777.f)); // poison: sparse scratch
    CorrParams p;
    p.in = in; p.in_stride = in_stride; p.out = out; p.out_stride = out_stride;
    p.corr = corr ? corr : h->scratch.data(); p.corr_stride = corr ? corr_stride : n; p.dense_corr = corr ? 1 : 0;
    p.hist_in = h->hist[h->cur].data(); p.hist_out = h->hist[h->cur ^ 1].data();
    p.Hpos = h->Hpos.data(); p.wtab = h->wtab.data();
    p.abits = h->abits.data(); p.abits_stride = astride;
    p.n = n; p.N = h->N; p.L = h->L; p.nseg = nseg; p.tiles_per_seg = tps; p.thresh = h->cs.thresh; p.corr_hist_zero = 0;
    emu_corr_main(&p, h->nchan, h->F);
    ResolveParams r;
    r.abits = p.abits; r.abits_stride = astride; r.L = h->L; r.corr = p.corr; r.corr_stride = p.corr_stride; r.dense_corr = p.dense_corr;
    r.in = in; r.in_stride = in_stride; r.hist_in = p.hist_in; r.corr_hist_zero = 0; r.taps = h->cs.symbols.data();
    r.n = n; r.N = h->N; r.isps = h->cs.isps; r.mark_delay = h->cs.mark_delay; r.written = h->written;
    r.emit_port1 = corr ? 1 : 0; r.tags = tags; r.tag_cap = tag_cap; r.tag_count = tag_count; r.atan_tab = aisx_atan_table;
    emu_corr_resolve(&r, h->nchan);
    h->cur ^= 1;
    h->written += (unsigned long long)n;
    return 0;
}

// ---- msk_timing_recovery_cc handle mirroring aisx_msk_* ----
struct EmuMsk {
    int nchan, osps;
    int lpw = 64;
    float d_sps, gain, gain_omega, limit;
    static constexpr int carry_cap = MSK_CARRY_MAX, ctag_cap = 64;
    std::vector<float> mu, omega;
    std::vector<int> div, carry_len[2], ctag_n[2], produced, consumed, status;
    std::vector<cf> dly1, dly2, diff1, tprev[2], carry[2], symscratch;
    std::vector<unsigned char> tbit[2];
    int tcur = 0;
    std::vector<unsigned long long> nread;
    std::vector<tag_rec> ctag[2];
    std::vector<msk_ctag> ct;
    std::vector<int> ct_n;
    int ct_cap = 0;
    int cur = 0;
    // time-parallel path (k_mskp.h): restart points per channel at most (-1: the serial kernel)
    int tp_smax = -1, tp_min_gap = 256, max_noutput = 0;
    int tp_join = 0; // 0: mskp_body<JOIN>, 1: the serial kernel with fast-forward (MskParams::ff)
    unsigned long long total_in = 0;
    long tp_stat[4] = { 0, 0, 0, 0 }; // restart points, units accepted, symbols taken from units, symbols in all
};

void* emu_msk_create(float sps, float gain, float limit, int osps, int nchan)
{
    EmuMsk* h = new EmuMsk();
    MskSetup ms = msk_setup(sps, gain);
    h->nchan = nchan; h->osps = osps; h->d_sps = ms.d_sps; h->gain = gain; h->gain_omega = ms.gain_omega; h->limit = limit;
    h->mu.assign(nchan, 0.5f); h->omega.assign(nchan, ms.d_sps); h->div.assign(nchan, 0);
    h->dly1.assign(nchan, mk(0, 0)); h->dly2 = h->dly1; h->diff1 = h->dly1;
    h->tprev[0] = h->dly1; h->tprev[1] = h->dly1;
    h->tbit[0].assign(nchan, 0); h->tbit[1].assign(nchan, 0); h->nread.assign(nchan, 0ull);
    for (int k = 0; k < 2; k++) {
        h->carry[k].assign((size_t)nchan * EmuMsk::carry_cap, mk(0, 0));
        h->carry_len[k].assign(nchan, 0);
        h->ctag[k].assign((size_t)nchan * EmuMsk::ctag_cap, tag_rec{ 0, 0, 0, 0 });
        h->ctag_n[k].assign(nchan, 0);
    }
    h->produced.assign(nchan, 0); h->consumed.assign(nchan, 0); h->status.assign(nchan, 0);
    return h;
}
void emu_msk_destroy(void* hv) { delete (EmuMsk*)hv; }
void emu_msk_set_lpw(void* hv, int lpw) { ((EmuMsk*)hv)->lpw = lpw; }
void emu_msk_set_time_parallel(void* hv, int smax, int min_gap, int max_noutput)
{
    EmuMsk* h = (EmuMsk*)hv;
    h->tp_smax = smax;
    h->tp_min_gap = min_gap;
    h->max_noutput = max_noutput;
}
void emu_msk_set_tp_join(void* hv, int which) { ((EmuMsk*)hv)->tp_join = which; }
void emu_msk_tp_stats(void* hv, long* out)
{
    for (int i = 0; i < 4; i++)
        out[i] = ((EmuMsk*)hv)->tp_stat[i];
}

static void emu_msk_fill(EmuMsk* h, MskParams& p);
static void emu_msk_tagprep(EmuMsk* h, const tag_rec* tags, const int* tag_counts, int tag_cap, int* ct_nc,
                            const msk_ctag* ctl_new = nullptr, const int* ctl_new_n = nullptr, int ctl_new_cap = 0);
// the time-parallel kernels on the lane model: prepass, units, join, gather
static void emu_mskp_run(EmuMsk* h, const cf* in, long in_stride, int n, const tag_rec* tags, const int* tag_counts,
                         int tag_cap, cf* syms, long out_stride, int* produced)
{
    const int nc = h->nchan;
    const bool ok = mskp_geometry_ok(h->d_sps, h->gain, h->limit, n + MSK_CARRY_MAX);
    const int smax = (ok && !(h->max_noutput > 0 && h->d_sps < 2.0f)) ? std::min(h->tp_smax, MSKP_SMAX) : 0;
    const int ctl_cap = MSKP_TPRE + (tags ? tag_cap : 0) + 1;
    std::vector<msk_ctag> ctl((size_t)nc * ctl_cap, msk_ctag{ 0, 0.f });
    std::vector<int> ctl_n(nc, 0), nrst(nc, 0), npieces(nc, 0);
    std::vector<mskp_rst> rst((size_t)nc * MSKP_SMAX);
    std::vector<mskp_res> res((size_t)nc * MSKP_SMAX);
    memset(res.data(), 0, res.size() * sizeof(mskp_res));
    std::vector<mskp_piece> pieces((size_t)nc * MSKP_SMAX);
    const long stage_stride = mskp_stage_stride(n, h->d_sps, h->gain, h->limit);
    std::vector<cf> stage((size_t)nc * stage_stride, mk(0, 0));
    MskpPrepParams pp;
    pp.nchan = nc; pp.tags = tags; pp.tag_count = tag_counts; pp.tag_cap = tag_cap; pp.W = h->total_in; pp.n = n;
    pp.d_sps = h->d_sps; pp.gain = h->gain; pp.limit = h->limit;
    pp.ctl = ctl.data(); pp.ctl_n = ctl_n.data(); pp.ctl_cap = ctl_cap;
    pp.smax = smax; pp.nrst = nrst.data(); pp.rst = rst.data(); pp.stage_stride = stage_stride;
    pp.tail = mskp_tail(h->d_sps); pp.min_gap = h->tp_min_gap;
    pp.max_span = getenv("AISX_MSK_TP_MAXSPAN") ? atoi(getenv("AISX_MSK_TP_MAXSPAN")) : 0x3fffffff;
    const bool sorted = !getenv("AISX_MSK_TP_UNSORTED");
    std::vector<int> ucount(8, 0), ulist((size_t)nc * MSKP_SMAX * MSKP_NCLS, 0);
    pp.ucount = sorted ? ucount.data() : nullptr; pp.ulist = ulist.data(); pp.ucap = (long)nc * MSKP_SMAX;
    run_grid(nc, 1, 64, MSKP_PREP_LDS_TAGS * 8, [&](EmuCtx& cx) { mskp_prep_body(cx, pp); });
    MskpParams p;
    p.nchan = nc; p.d_sps = h->d_sps; p.gain = h->gain; p.gain_omega = h->gain_omega; p.limit = h->limit;
    p.mu = h->mu.data(); p.omega = h->omega.data(); p.div = h->div.data();
    p.dly1 = h->dly1.data(); p.dly2 = h->dly2.data(); p.diff1 = h->diff1.data(); p.nread = h->nread.data();
    p.in = in; p.in_stride = in_stride; p.n = n;
    p.carry_in = h->carry[h->cur].data(); p.carry_out = h->carry[h->cur ^ 1].data();
    p.carry_len_in = h->carry_len[h->cur].data(); p.carry_len_out = h->carry_len[h->cur ^ 1].data(); p.carry_cap = EmuMsk::carry_cap;
    p.ctag_in = h->ctag[h->cur].data(); p.ctag_n_in = h->ctag_n[h->cur].data();
    p.ctag_out = h->ctag[h->cur ^ 1].data(); p.ctag_n_out = h->ctag_n[h->cur ^ 1].data(); p.ctag_cap = EmuMsk::ctag_cap;
    p.ctl = ctl.data(); p.ctl_n = ctl_n.data(); p.ctl_cap = ctl_cap;
    p.smax = std::max(smax, 1); p.nrst = nrst.data(); p.rst = rst.data(); p.res = res.data();
    p.stage = stage.data(); p.stage_stride = stage_stride;
    p.syms = syms; p.out_stride = out_stride; p.out_cap = (int)out_stride;
    p.pieces = pieces.data(); p.npieces = npieces.data();
    p.produced = produced; p.consumed = h->consumed.data(); p.status = h->status.data();
    p.mmse = &aisx_mmse_taps[0][0];
    p.W = h->total_in; p.look = mskp_look(h->d_sps, h->limit); p.padv = mskp_padv(h->d_sps, h->gain, h->limit); p.padv_inv = mskp_padv_inv(h->d_sps, h->gain, h->limit); p.jw = getenv("AISX_MSK_JW") ? atoi(getenv("AISX_MSK_JW")) : 16; p.tail = pp.tail; p.max_noutput = h->max_noutput;
    p.ucount = pp.ucount; p.ulist = pp.ulist; p.ucap = pp.ucap;
    if (smax > 0)
        run_grid((nc * smax + 63) / 64 + (sorted ? MSKP_NCLS : 0), 1, 64, MSKP_LDS_BYTES, [&](EmuCtx& cx) { mskp_body<EmuCtx, false>(cx, p); });
    if (h->tp_join == 0) {
        run_grid((nc + p.jw - 1) / p.jw, 1, 64, MSKP_LDS_BYTES, [&](EmuCtx& cx) { mskp_body<EmuCtx, true>(cx, p); });
    } else { // the serial kernel as the join (MskParams::ff)
        std::vector<int> ct_nc(nc, 0);
        emu_msk_tagprep(h, tags, tag_counts, tag_cap, ct_nc.data(), ctl.data(), ctl_n.data(), ctl_cap); // (new tags as the prepass left them)
        MskParams m;
        emu_msk_fill(h, m);
        m.in = in; m.in_stride = in_stride; m.n = n; m.stream_mode = 1; m.gr_ninput = 0; m.gr_noutput = 0;
        m.syms = syms; m.err = nullptr; m.mu_out = nullptr; m.out_stride = out_stride; m.out_cap = (int)out_stride;
        m.sym_al16 = 0; m.produced = produced; m.inline_tags = 0;
        m.ff = 1; m.nrst = nrst.data(); m.rst = rst.data(); m.res = res.data(); m.pieces = pieces.data(); m.npieces = npieces.data();
        m.ct_nc = ct_nc.data();
        emu_msk(&m);
    }
    MskpGatherParams g;
    g.nchan = nc; g.pieces = pieces.data(); g.npieces = npieces.data(); g.stage = stage.data(); g.stage_stride = stage_stride;
    g.syms = syms; g.out_stride = out_stride;
    run_grid(MSKP_GATHER_X, nc, 64, 64, [&](EmuCtx& cx) { mskp_gather_body(cx, g); });
    for (int c = 0; c < nc; c++) {
        h->tp_stat[0] += nrst[c];
        h->tp_stat[1] += npieces[c];
        for (int i = 0; i < npieces[c]; i++)
            h->tp_stat[2] += pieces[(size_t)c * MSKP_SMAX + i].cnt;
        h->tp_stat[3] += produced[c];
    }
}

static void emu_msk_fill(EmuMsk* h, MskParams& p)
{
    p.nchan = h->nchan; p.d_sps = h->d_sps; p.gain = h->gain; p.gain_omega = h->gain_omega; p.limit = h->limit; p.osps = h->osps;
    p.mu = h->mu.data(); p.omega = h->omega.data(); p.div = h->div.data();
    p.dly1 = h->dly1.data(); p.dly2 = h->dly2.data(); p.diff1 = h->diff1.data();
    p.nread = h->nread.data();
    p.carry_in = h->carry[h->cur].data(); p.carry_out = h->carry[h->cur ^ 1].data();
    p.carry_len_in = h->carry_len[h->cur].data(); p.carry_len_out = h->carry_len[h->cur ^ 1].data(); p.carry_cap = EmuMsk::carry_cap;
    p.ctag_out = h->ctag[h->cur ^ 1].data();
    p.ctag_n_out = h->ctag_n[h->cur ^ 1].data(); p.ctag_cap = EmuMsk::ctag_cap;
    p.ct = h->ct.data(); p.ct_n = h->ct_n.data(); p.ct_cap = h->ct_cap;
    p.consumed = h->consumed.data(); p.status = h->status.data();
    p.mmse = &aisx_mmse_taps[0][0];
    // free-running lanes: every lane gets a tag-queue column of its own (see MskParams)
    p.lpw = h->lpw;
    p.lds_wave_stride = msk_lds_ring(h->lpw) + MSK_TAGQ * 64 * 8 + h->lpw * 8;
    p.lds_ring_off = msk_lds_ringoff(h->lpw);
    p.tq_stride = 64;
    p.tq_private = 1;
    p.inline_tags = getenv("AISX_MSK_INLINE_TAGS") ? atoi(getenv("AISX_MSK_INLINE_TAGS")) : 1;
    p.max_noutput = h->max_noutput;
    p.ff = 0; p.nrst = nullptr; p.rst = nullptr; p.res = nullptr; p.pieces = nullptr; p.npieces = nullptr; p.ct_nc = nullptr;
    p.lds_tab_off = p.lds_ring_off + msk_waves(h->lpw) * p.lds_wave_stride;
}

static void emu_msk_tagprep(EmuMsk* h, const tag_rec* tags, const int* tag_counts, int tag_cap, int* ct_nc,
                            const msk_ctag* ctl_new, const int* ctl_new_n, int ctl_new_cap)
{
    h->ct_cap = EmuMsk::ctag_cap + (tags ? tag_cap : 0);
    h->ct.assign((size_t)h->nchan * h->ct_cap, msk_ctag{ 0, 0.f });
    h->ct_n.assign(h->nchan, 0);
    TagPrepParams t;
    t.nchan = h->nchan;
    t.ctag_in = h->ctag[h->cur].data(); t.ctag_n_in = h->ctag_n[h->cur].data(); t.ctag_cap = EmuMsk::ctag_cap;
    t.tags = tags; t.tag_count = tag_counts; t.tag_cap = tag_cap;
    t.nread = h->nread.data();
    t.ct = h->ct.data(); t.ct_n = h->ct_n.data(); t.ct_cap = h->ct_cap; t.ct_nc = ct_nc;
    t.ctl_new = ctl_new; t.ctl_new_n = ctl_new_n; t.ctl_new_cap = ctl_new_cap; t.ctl_new_pre = MSKP_TPRE; t.W = h->total_in;
    run_grid((h->nchan + 3) / 4, 1, 256, 64, [&](EmuCtx& cx) { tagprep_body(cx, t); });
}

static void emu_msk_bittail(EmuMsk* h, const cf* syms, long sym_stride, const int* produced, unsigned char* bits,
                            long bit_stride, int max_out)
{
    BitTailParams b;
    b.nchan = h->nchan; b.syms = syms; b.sym_stride = sym_stride; b.produced = produced;
    b.bits = bits; b.bit_stride = bit_stride;
    b.prev_sym_in = h->tprev[h->tcur].data(); b.prev_bit_in = h->tbit[h->tcur].data();
    b.prev_sym_out = h->tprev[h->tcur ^ 1].data(); b.prev_bit_out = h->tbit[h->tcur ^ 1].data();
    b.atan_tab = aisx_atan_table;
    emu_bittail(&b, max_out);
    h->tcur ^= 1;
}

int emu_msk_process_stream(void* hv, const cf* in, long in_stride, int n, const tag_rec* tags, const int* tag_counts,
                           int tag_cap, cf* syms, float* err, float* mu, unsigned char* bits, long out_stride,
                           int* produced, int* consumed_out)
{
    EmuMsk* h = (EmuMsk*)hv;
    if (!syms) {
        h->symscratch.resize((size_t)h->nchan * out_stride);
        syms = h->symscratch.data();
    }
    if (h->tp_smax >= 0 && h->osps == 1 && !err && !mu) {
        emu_mskp_run(h, in, in_stride, n, tags, tag_counts, tag_cap, syms, out_stride, produced);
    } else {
        emu_msk_tagprep(h, tags, tag_counts, tag_cap, nullptr);
        MskParams p;
        emu_msk_fill(h, p);
        p.in = in; p.in_stride = in_stride; p.n = n; p.stream_mode = 1; p.gr_ninput = 0; p.gr_noutput = 0;
        p.syms = syms; p.err = err; p.mu_out = mu; p.out_stride = out_stride; p.out_cap = (int)out_stride;
        p.sym_al16 = ((uintptr_t)syms % 16 == 0) && (out_stride % 2 == 0);
        p.produced = produced;
        emu_msk(&p);
    }
    h->total_in += (unsigned long long)n;
    h->cur ^= 1;
    if (bits)
        emu_msk_bittail(h, syms, out_stride, produced, bits, out_stride, (int)out_stride);
    int st = 0;
    for (int c = 0; c < h->nchan; c++) {
        st |= h->status[c];
        if (consumed_out)
            consumed_out[c] = h->consumed[c];
    }
    return st;
}

int emu_msk_general_work(void* hv, int noutput, int ninput, const cf* in /* in[ninput] readable */, cf* out,
                         float* err, float* mu, unsigned char* bits, const tag_rec* tags, int ntags,
                         unsigned long long nitems_read, int* consumed, int* produced)
{
    EmuMsk* h = (EmuMsk*)hv;
    h->nread[0] = nitems_read;
    h->carry_len[h->cur][0] = 0;
    h->ctag_n[h->cur][0] = 0;
    emu_msk_tagprep(h, tags, &ntags, ntags + 1, nullptr);
    MskParams p;
    emu_msk_fill(h, p);
    p.in = in; p.in_stride = ninput + 1; p.n = ninput; p.stream_mode = 0; p.gr_ninput = ninput; p.gr_noutput = noutput;
    p.syms = out; p.err = err; p.mu_out = mu; p.out_stride = noutput; p.out_cap = noutput;
    p.sym_al16 = ((uintptr_t)out % 16 == 0) && (noutput % 2 == 0);
    p.produced = h->produced.data();
    emu_msk(&p);
    h->cur ^= 1;
    if (bits)
        emu_msk_bittail(h, out, noutput, h->produced.data(), bits, noutput, noutput);
    *consumed = h->consumed[0];
    *produced = h->produced[0];
    return h->status[0];
}

void emu_pfb(const PfbParams* p, int nstreams)
{
    run_grid((p->nframes + 3) / 4, nstreams, PFB_T, PFB_LDS_BYTES, [&](EmuCtx& cx) { pfb_body(cx, *p); });
}
struct EmuPfb {
    int nstreams, D, K, Lh;
    std::vector<float> taps;
    std::vector<cf> wtab, hist[2];
    int cur = 0;
    long frame0 = 0;
};
void* emu_pfb_create(int decim, const float* taps, int ntaps, int nstreams)
{
    EmuPfb* h = new EmuPfb();
    h->nstreams = nstreams; h->D = decim; h->K = (ntaps + PFB_M - 1) / PFB_M; h->Lh = h->K * PFB_M;
    h->taps.assign(h->Lh, 0.f);
    for (int i = 0; i < ntaps; i++) h->taps[i] = taps[i];
    h->wtab.resize(PFB_M);
    for (int k = 0; k < PFB_M; k++) { double a = -2.0 * M_PI * k / PFB_M; h->wtab[k] = mk((float)cos(a), (float)sin(a)); }
    h->hist[0].assign((size_t)nstreams * h->Lh, mk(0, 0)); h->hist[1] = h->hist[0];
    return h;
}
void emu_pfb_destroy(void* hv) { delete (EmuPfb*)hv; }
int emu_pfb_process(void* hv, const cf* in, long in_stride, int n, cf* out, long out_stride)
{
    EmuPfb* h = (EmuPfb*)hv;
    PfbParams p;
    p.in = in; p.in_stride = in_stride; p.hist_in = h->hist[h->cur].data(); p.hist_out = h->hist[h->cur ^ 1].data();
    p.taps = h->taps.data(); p.wtab = h->wtab.data(); p.out = out; p.out_stride = out_stride;
    p.n = n; p.D = h->D; p.K = h->K; p.Lh = h->Lh; p.nframes = n / h->D; p.frame0 = h->frame0;
    emu_pfb(&p, h->nstreams);
    h->cur ^= 1; h->frame0 += p.nframes;
    return p.nframes;
}
float emu_fast_atan2f(float y, float x) { return fast_atan2f_tab(y, x, aisx_atan_table); }
void emu_fxpt_float_to_fixed_n(const float* x, int* out, long n)
{
    for (long i = 0; i < n; i++) out[i] = fxpt_float_to_fixed(x[i]);
}
// k_agcw.h's float_to_fixed of an NCO phase against the general statement
void emu_nco_phase_to_fixed_n(const float* x, int* out, long n)
{
    for (long i = 0; i < n; i++) out[i] = nco_phase_to_fixed(x[i]);
}
const float* emu_mmse_table() { return &aisx_mmse_taps[0][0]; }
const float* emu_atan_table() { return aisx_atan_table; }

#ifdef HAVE_AGC
struct EmuAgc {
    int nchan, W;
    float ref;
    std::vector<cf> hist[2];
    int cur = 0;
    bool no_stream = false; // tile kernels only (agc8_body / agc_body), as for windows agcw_body does not serve
};
void emu_agc_set_streaming(void* hv, int on) { ((struct EmuAgc*)hv)->no_stream = !on; }
void* emu_agc_create(int nsamples, float reference, int nchan)
{
    EmuAgc* h = new EmuAgc();
    h->nchan = nchan; h->W = nsamples; h->ref = reference;
    h->hist[0].assign((size_t)nchan * nsamples, mk(0, 0));
    h->hist[1] = h->hist[0];
    return h;
}
void emu_agc_destroy(void* hv) { delete (EmuAgc*)hv; }
void emu_agc_process(void* hv, const cf* in, long in_stride, cf* out, long out_stride, int n)
{
    EmuAgc* h = (EmuAgc*)hv;
    AgcParams p;
    p.in = in; p.in_stride = in_stride; p.out = out; p.out_stride = out_stride;
    p.hist_in = h->hist[h->cur].data(); p.hist_out = h->hist[h->cur ^ 1].data();
    p.n = n; p.W = h->W; p.reference = h->ref; p.floor_env = AGC_FLOOR_DEFAULT; p.ntiles = agc8_applies(h->W) ? (n + AGC8_TL - 1) / AGC8_TL : (n + AGC_TL - 1) / AGC_TL;
    p.phases = nullptr; p.phases_stride = 0; p.dvec = nullptr; p.dvec_stride = 0; p.sintab = nullptr; p.pend_in = nullptr; p.pend_out = nullptr; p.npend = 0; p.n_raw = 0;
    if (agcw_applies(p.W, n) && !h->no_stream) // (the product's dispatch: aisx_agc_process)
        run_grid(agcw_grid(n), h->nchan, AGW_T, AGW_LDS_BYTES, [&](EmuCtx& cx) { agcw_body<false>(cx, p); });
    else if (agc8_applies(p.W))
        run_grid(p.ntiles, h->nchan, AGC8_T, AGC8_LDS_BYTES, [&](EmuCtx& cx) { agc8_body(cx, p); });
    else
        run_grid(p.ntiles, h->nchan, AGC_T, AGC_LDS_BYTES, [&](EmuCtx& cx) { agc_body(cx, p); });
    h->cur ^= 1;
}
#endif
#ifdef HAVE_FREQSYNC
struct EmuFs {
    int nchan, offset, max_vec;
    float binsize, sens;
    std::vector<cf> pend[2], wtab;
    std::vector<int> maxpos;
    std::vector<float> phase;
    int cur = 0, npend = 0;
};
void* emu_fs_create(double samplerate, double bits_per_sec, int fftlen, int nchan, int max_items)
{
    if (fftlen != FS_F)
        return nullptr;
    EmuFs* h = new EmuFs();
    h->nchan = nchan;
    const float sr = (float)(int)samplerate;
    const int dr = (int)bits_per_sec;
    h->offset = (int)(fftlen * ((float)dr / sr));
    h->binsize = sr / (float)fftlen;
    h->sens = (float)(-1.0 / (samplerate / (2 * M_PI)));
    h->max_vec = (max_items + fftlen) / fftlen + 1;
    h->pend[0].assign((size_t)nchan * FS_F, mk(0, 0));
    h->pend[1] = h->pend[0];
    h->wtab.resize(FS_F);
    for (int k = 0; k < FS_F; k++) {
        double a = -2.0 * M_PI * (double)k / (double)FS_F;
        h->wtab[k] = mk((float)cos(a), (float)sin(a));
    }
    h->maxpos.assign((size_t)nchan * h->max_vec, 0);
    h->phase.assign(nchan, 0.f);
    return h;
}
void emu_fs_destroy(void* hv) { delete (EmuFs*)hv; }
int emu_fs_process(void* hv, const cf* in, long in_stride, int n, cf* out, long out_stride, float* fhat, long fhat_stride)
{
    EmuFs* h = (EmuFs*)hv;
    const int nvec = (h->npend + n) / FS_F;
    if (nvec > 0) {
        FsEstParams e;
        e.in = in; e.in_stride = in_stride; e.pend = h->pend[h->cur].data(); e.npend = h->npend; e.wtab = h->wtab.data();
        e.maxpos = h->maxpos.data(); e.maxpos_stride = h->max_vec; e.nvec = nvec; e.offset = h->offset;
        run_grid((nvec + FS_VEC_PER_WG - 1) / FS_VEC_PER_WG, h->nchan, FS_T, FS_LDS_BYTES, [&](EmuCtx& cx) { fs_est_body(cx, e); });
    }
    FsMixParams m;
    m.nchan = h->nchan; m.in = in; m.in_stride = in_stride; m.pend_in = h->pend[h->cur].data(); m.pend_out = h->pend[h->cur ^ 1].data();
    m.npend = h->npend; m.n = n; m.out = out; m.out_stride = out_stride; m.maxpos = h->maxpos.data(); m.maxpos_stride = h->max_vec;
    m.fhat = fhat; m.fhat_stride = fhat_stride; m.phase = h->phase.data(); m.nvec = nvec; m.binsize = h->binsize; m.sensitivity = h->sens; m.sintab = &aisx_sine_table[0][0];
    run_grid((h->nchan + FSM_CPW - 1) / FSM_CPW, 1, FSM_T, FSM_LDS_BYTES, [&](EmuCtx& cx) { fs_mix_body(cx, m); });
    h->npend = h->npend + n - nvec * FS_F;
    h->cur ^= 1;
    return nvec * FS_F;
}
// the fused front end (aisx_freqsync_agc_process): estimates, phase walk, mixing inside the AGC
int emu_fs_agc_process(void* fv, void* av, const cf* in, long in_stride, int n, cf* out, long out_stride, float* fhat, long fhat_stride)
{
    EmuFs* h = (EmuFs*)fv;
    EmuAgc* a = (EmuAgc*)av;
    const int nvec = (h->npend + n) / FS_F, total = nvec * FS_F;
    const long pstride = ((long)h->max_vec * (FS_F / FSW_CK) + 3) & ~3L;
    std::vector<float> phases((size_t)h->nchan * pstride), dvec((size_t)h->nchan * h->max_vec);
    if (nvec > 0) {
        FsEstParams e;
        e.in = in; e.in_stride = in_stride; e.pend = h->pend[h->cur].data(); e.npend = h->npend; e.wtab = h->wtab.data();
        e.maxpos = h->maxpos.data(); e.maxpos_stride = h->max_vec; e.nvec = nvec; e.offset = h->offset;
        run_grid((nvec + FS_VEC_PER_WG - 1) / FS_VEC_PER_WG, h->nchan, FS_T, FS_LDS_BYTES, [&](EmuCtx& cx) { fs_est_body(cx, e); });
        FsWalkParams w;
        w.nchan = h->nchan; w.maxpos = h->maxpos.data(); w.maxpos_stride = h->max_vec; w.fhat = fhat; w.fhat_stride = fhat_stride;
        w.phase_in = h->phase.data(); w.phase_out = h->phase.data(); w.phases = phases.data(); w.phases_stride = pstride; w.dvec = dvec.data(); w.dvec_stride = h->max_vec; w.nvec = nvec; w.binsize = h->binsize;
        w.sensitivity = h->sens;
        run_grid((h->nchan + FSW_T - 1) / FSW_T, 1, FSW_T, FSW_LDS_BYTES, [&](EmuCtx& cx) { fs_walk_body(cx, w); });
    }
    AgcParams p;
    p.in = in; p.in_stride = in_stride; p.out = out; p.out_stride = out_stride;
    p.hist_in = a->hist[a->cur].data(); p.hist_out = a->hist[a->cur ^ 1].data();
    p.n = total; p.W = a->W; p.reference = a->ref; p.floor_env = AGC_FLOOR_DEFAULT;
    p.ntiles = total > 0 ? (total + AGC8_TL - 1) / AGC8_TL : 1;
    p.phases = phases.data(); p.phases_stride = pstride; p.dvec = dvec.data(); p.dvec_stride = h->max_vec; p.sintab = &aisx_sine_table[0][0]; p.pend_in = h->pend[h->cur].data(); p.pend_out = h->pend[h->cur ^ 1].data();
    p.npend = h->npend; p.n_raw = n;
    if (agcw_applies(p.W, total) && !a->no_stream) // (the product's dispatch: aisx_freqsync_agc_process)
        run_grid(agcw_grid(total), h->nchan, AGW_T, AGW_LDS_BYTES, [&](EmuCtx& cx) { agcw_body<true>(cx, p); });
    else
        run_grid(p.ntiles, h->nchan, AGC8_T, AGC8_LDS_BYTES_MIXED, [&](EmuCtx& cx) { agc8_body(cx, p); });
    h->npend = h->npend + n - total;
    h->cur ^= 1;
    a->cur ^= 1;
    return total;
}
void emu_freqest_work(void* hv, const cf* vecs, long vec_stride, float* out, long out_stride, int nvec)
{
    EmuFs* h = (EmuFs*)hv;
    FsFreqestParams p;
    p.vecs = vecs; p.vec_stride = vec_stride; p.out = out; p.out_stride = out_stride; p.nvec = nvec; p.fftlen = FS_F;
    p.offset = h->offset; p.binsize = h->binsize;
    run_grid(h->nchan, 1, 64, 0, [&](EmuCtx& cx) { fs_freqest_body(cx, p); });
}
// the block alone for any vector length (aisx_freqest_create_n: offset and bin size from the float rate)
void emu_freqest_any(const cf* vecs, long vec_stride, float* out, long out_stride, int nvec, int nchan, int fftlen,
                     float sample_rate, int data_rate)
{
    FsFreqestParams p;
    p.vecs = vecs; p.vec_stride = vec_stride; p.out = out; p.out_stride = out_stride; p.nvec = nvec; p.fftlen = fftlen;
    p.offset = (int)(fftlen * ((float)data_rate / sample_rate));
    p.binsize = sample_rate / (float)fftlen;
    run_grid(nchan, 1, 64, 0, [&](EmuCtx& cx) { fs_freqest_body(cx, p); });
}
#endif
}
