// aisx_devctx.h -- the gfx950 execution context the kernel bodies run under in
// the product: wave64 ballot/shuffle builtins, LDS, workgroup barrier, global
// atomics.  (tests/emul has the CPU model of the same interface.)
#pragma once
#include <hip/hip_runtime.h>
#include "aisx_common.h"

namespace aisx {

struct DevCtx {
    // LDS writes of one lane are seen by the other lanes of its wave at once (they run in lock step)
    static constexpr bool wave_lds_coherent = true;
    char* lds_;
    __device__ __forceinline__ int tid() const { return threadIdx.x; }
    __device__ __forceinline__ int nthreads() const { return blockDim.x; }
    __device__ __forceinline__ int bx() const { return blockIdx.x; }
    __device__ __forceinline__ int by() const { return blockIdx.y; }
    __device__ __forceinline__ char* lds() const { return lds_; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    // lanes of a wave run in lock step and LDS operations of a wave complete in order: nothing
    // to do (the CPU lane model needs a real barrier here)
    __device__ __forceinline__ void wave_sync() const { __builtin_amdgcn_wave_barrier(); }
    // an LDS exchange among the lanes of ONE wave: the wave's LDS instructions are carried out in
    // the order it issued them, so a read issued behind a write sees it; only the compiler has to
    // be kept from moving one across the other
    __device__ __forceinline__ void wave_lds_sync() const
    {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    // a lane that leaves the kernel for good before its workgroup's barriers (nothing to do
    // on the device: the hardware counts waves, not lanes)
    __device__ __forceinline__ void retire() const {}
    __device__ __forceinline__ unsigned long long ballot(bool p) const { return __builtin_amdgcn_ballot_w64(p); }
    // per-lane predicate from a wave-uniform mask (the mask goes straight into exec / vcc)
    __device__ __forceinline__ bool inv_ballot(unsigned long long m) const { return __builtin_amdgcn_inverse_ballot_w64(m); }
    // complex primitives of the FFT butterflies (k_fft.h): one packed-fp32 instruction each,
    // the +-j rotation done by the operand-select / negate modifiers
    typedef float v2f __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ v2f tov(cf a) { v2f r; r.x = a.re; r.y = a.im; return r; }
    static __device__ __forceinline__ cf tocf(v2f a) { return mk(a.x, a.y); }
    __device__ __forceinline__ cf cadd(cf a, cf b) const { return tocf(tov(a) + tov(b)); }
    __device__ __forceinline__ cf csub(cf a, cf b) const { return tocf(tov(a) - tov(b)); }
    __device__ __forceinline__ cf add_mj(cf a, cf b) const // a - j b = (a.re + b.im, a.im - b.re)
    {
        v2f r;
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(tov(a)), "v"(tov(b)));
        return tocf(r);
    }
    __device__ __forceinline__ cf add_pj(cf a, cf b) const // a + j b = (a.re - b.im, a.im + b.re)
    {
        v2f r;
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(tov(a)), "v"(tov(b)));
        return tocf(r);
    }
    // a (c + j s) with c = +-K[CS], s = +-K[SS]; K = 0: (cos(pi/8), sin(pi/8)), 1: (sqrt(1/2), sqrt(1/2)),
    // each living in one scalar register pair for the whole kernel
    template <int KSEL, int CS, bool CNEG, int SS, bool SNEG>
    __device__ __forceinline__ cf cmul_sel(cf a) const
    {
        v2f k;
        if (KSEL == 0) {
            k.x = 0.92387953251128673848f;
            k.y = 0.38268343236508978178f;
        } else {
            k.x = 0.70710678118654752440f;
            k.y = 0.70710678118654752440f;
        }
        v2f m, r;
        const v2f av = tov(a);
        // m = (a.re c, a.im c)
        if (CS == 0 && !CNEG)
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(m) : "v"(av), "s"(k));
        else if (CS == 0 && CNEG)
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(m) : "v"(av), "s"(k));
        else if (CS == 1 && !CNEG)
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(m) : "v"(av), "s"(k));
        else
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(m) : "v"(av), "s"(k));
        // r = (-a.im s + m.re, a.re s + m.im)
        if (SS == 0 && !SNEG)
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(av), "s"(k), "v"(m));
        else if (SS == 0 && SNEG)
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(av), "s"(k), "v"(m));
        else if (SS == 1 && !SNEG)
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(av), "s"(k), "v"(m));
        else
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(av), "s"(k), "v"(m));
        return tocf(r);
    }
    // a b and a conj(b), rounded as cmul_fma / cmul_conj_fma (aisx_common.h): two packed instructions, the
    // cross terms picked by operand-select and negate modifiers.  (Left to the compiler, each product with a
    // loop-invariant b keeps a swapped and a negated copy of b in registers next to b itself: k_corr4e.h has 29
    // such constants per thread and 128 VGPRs.)
    __device__ __forceinline__ cf cmul(cf a, cf b) const
    {
        v2f m, r;
        const v2f av = tov(a), bv = tov(b);
        // m = (a.re b.re, a.re b.im);  r = (-a.im b.im + m.re, a.im b.re + m.im)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(m) : "v"(av), "v"(bv));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(av), "v"(bv), "v"(m));
        return tocf(r);
    }
    __device__ __forceinline__ cf cmul_conj(cf a, cf b) const
    {
        v2f m, r;
        const v2f av = tov(a), bv = tov(b);
        // m = (a.re b.re, -(a.re b.im));  r = (a.im b.im + m.re, a.im b.re + m.im)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(m) : "v"(av), "v"(bv));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(av), "v"(bv), "v"(m));
        return tocf(r);
    }
    // lo = v of lane (i & ~W), hi = v of lane (i | W), W = 16 or 32: one row swap per 32 bits
    // (v_permlane16_swap / v_permlane32_swap with both operands = v)
    template <int W>
    __device__ __forceinline__ void pair_rows(cf v, cf& lo, cf& hi) const
    {
        static_assert(W == 16 || W == 32, "row width");
        const unsigned a = __float_as_uint(v.re), b = __float_as_uint(v.im);
        if (W == 16) {
            const auto ra = __builtin_amdgcn_permlane16_swap(a, a, false, false);
            const auto rb = __builtin_amdgcn_permlane16_swap(b, b, false, false);
            lo = mk(__uint_as_float(ra[0]), __uint_as_float(rb[0]));
            hi = mk(__uint_as_float(ra[1]), __uint_as_float(rb[1]));
        } else {
            const auto ra = __builtin_amdgcn_permlane32_swap(a, a, false, false);
            const auto rb = __builtin_amdgcn_permlane32_swap(b, b, false, false);
            lo = mk(__uint_as_float(ra[0]), __uint_as_float(rb[0]));
            hi = mk(__uint_as_float(ra[1]), __uint_as_float(rb[1]));
        }
    }
    // Transpose of eight complex registers against lane bits 3 .. 5: x[j] of lane (h, e) <- x[h] of lane (j, e), h / j = bits
    // 5:3 of the lane, e = bits 2:0.  Three stages of pairwise half exchanges: lane bit 5 against register bit 2 by
    // v_permlane32_swap (lanes 32-63 of the first operand with lanes 0-31 of the second), lane bit 4 against register bit 1
    // by v_permlane16_swap (odd rows of 16 with even rows), lane bit 3 against register bit 0 by two DPP moves per dword
    // (row_ror:8 = the other half of the row, written under a bank mask): 32 VALU instructions for what an exchange through
    // LDS does with 8 ds_write_b64 + 8 ds_read_b64 and a wait.
    __device__ __forceinline__ void xpose8_lane_hi(cf (&x)[8]) const
    {
        unsigned u[16];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            u[2 * k] = __float_as_uint(x[k].re);
            u[2 * k + 1] = __float_as_uint(x[k].im);
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const auto q = __builtin_amdgcn_permlane32_swap(u[2 * r + d], u[2 * (r + 4) + d], false, false);
                u[2 * r + d] = q[0];
                u[2 * (r + 4) + d] = q[1];
            }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int r = (rr & 1) | ((rr & 2) << 1); // 0, 1, 4, 5
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const auto q = __builtin_amdgcn_permlane16_swap(u[2 * r + d], u[2 * (r + 2) + d], false, false);
                u[2 * r + d] = q[0];
                u[2 * (r + 2) + d] = q[1];
            }
        }
#pragma unroll
        for (int r = 0; r < 8; r += 2)
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const int a = (int)u[2 * r + d], b = (int)u[2 * (r + 1) + d];
                // lanes 0-7 of each row: b <- a of lane + 8; lanes 8-15: a <- b of lane - 8
                const int nb = __builtin_amdgcn_update_dpp(b, a, 0x128, 0xf, 0x3, false);
                const int na = __builtin_amdgcn_update_dpp(a, b, 0x128, 0xf, 0xc, false);
                u[2 * r + d] = (unsigned)na;
                u[2 * (r + 1) + d] = (unsigned)nb;
            }
#pragma unroll
        for (int k = 0; k < 8; k++)
            x[k] = mk(__uint_as_float(u[2 * k]), __uint_as_float(u[2 * k + 1]));
    }
    // v of lane - 1 / lane + 1, 0 at the ends of the wave: one DPP move each (wave_shr:1 / wave_shl:1)
    __device__ __forceinline__ unsigned lane_prev_u32(unsigned v) const { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); }
    __device__ __forceinline__ unsigned lane_next_u32(unsigned v) const { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false); }
    // Wave scans of non-negative integers (k_agcw.h: bit patterns of envelopes), 0 the identity.
    // excl_prefix: max over the lanes below this one; excl_suffix: over the lanes above it.
    // DPP row shifts fill with 0 where a lane has no source (bound_ctrl), which is the identity here.
    template <int CTRL, int ROWMASK>
    static __device__ __forceinline__ int dpp_max_nn(int v)
    {
        return max(v, __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xf, true));
    }
    __device__ __forceinline__ int wave_excl_prefix_max_nn(int v) const
    {
        v = dpp_max_nn<0x111, 0xf>(v); // row_shr:1
        v = dpp_max_nn<0x112, 0xf>(v); // row_shr:2
        v = dpp_max_nn<0x114, 0xf>(v); // row_shr:4
        v = dpp_max_nn<0x118, 0xf>(v); // row_shr:8   -> inclusive inside each row of 16
        v = dpp_max_nn<0x142, 0xa>(v); // row_bcast:15 into rows 1 and 3
        v = dpp_max_nn<0x143, 0xc>(v); // row_bcast:31 into rows 2 and 3 -> inclusive over the wave
        return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); // wave_shr:1, lane 0 gets 0
    }
    // (`all`: the maximum over the whole wave, wave-uniform)
    __device__ __forceinline__ int wave_excl_suffix_max_nn(int v, int& all) const
    {
        v = dpp_max_nn<0x101, 0xf>(v); // row_shl:1
        v = dpp_max_nn<0x102, 0xf>(v); // row_shl:2
        v = dpp_max_nn<0x104, 0xf>(v); // row_shl:4
        v = dpp_max_nn<0x108, 0xf>(v); // row_shl:8   -> inclusive inside each row; lane 16 r holds row r's maximum
        // (no broadcast runs towards lower rows: the row maxima go through scalar registers)
        const int t0 = __builtin_amdgcn_readlane(v, 0), t1 = __builtin_amdgcn_readlane(v, 16);
        const int t2 = __builtin_amdgcn_readlane(v, 32), t3 = __builtin_amdgcn_readlane(v, 48);
        const int T2 = max(t2, t3), T1 = max(t1, T2);
        all = max(t0, T1);
        const int row = (int)(threadIdx.x & 63u) >> 4;
        const int above = row == 0 ? T1 : row == 1 ? T2 : row == 2 ? t3 : 0;
        v = max(v, above);
        return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); // wave_shl:1, lane 63 gets 0
    }
    // v_rcp_f32: the reciprocal to 1 ulp (callers refine it, k_agcw.h: agcw_gain_fast)
    __device__ __forceinline__ float rcp_approx(float x) const { return __builtin_amdgcn_rcpf(x); }
    // v_sqrt_f32: the square root to 1 ulp (k_freqsync.h: scores that only select what is evaluated exactly)
    __device__ __forceinline__ float sqrt_approx(float x) const { return __builtin_amdgcn_sqrtf(x); }
    // entry `idx` of a table of 8-byte pairs in LDS (k_agcw.h: the NCO's sine table)
    __device__ __forceinline__ cf lds_cf(const float* tab, unsigned idx) const { return ld8(reinterpret_cast<const cf*>(tab) + idx); }
    // x - floorf(x) for x >= 0 (v_fract_f32 is exact there)
    __device__ __forceinline__ float fract(float x) const { return __builtin_amdgcn_fractf(x); }
    // two complex items to byte offset `off` (< 4 GiB) of `base`: one 16-byte store
    __device__ __forceinline__ void store16(cf* base, unsigned off, cf a, cf b) const
    {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v;
        v.x = a.re;
        v.y = a.im;
        v.z = b.re;
        v.w = b.im;
        *(f4*)((char*)base + off) = v;
    }
    // lane-wise select by a wave mask, as one v_cndmask_b32 the optimiser cannot look into
    __device__ __forceinline__ float sel_f32(unsigned long long m, float if_set, float if_clear) const
    {
        float r;
        asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
        return r;
    }
    // scheduling fences: the value is materialised here, in program order with the other pins
    // (keeps speculated arithmetic above the branch that may discard it)
    __device__ __forceinline__ void pin(float& v) const { asm volatile("" : "+v"(v)); }
    __device__ __forceinline__ void pin(int& v) const { asm volatile("" : "+v"(v)); }
    __device__ __forceinline__ void pin_mask(unsigned long long& m) const { asm volatile("" : "+s"(m)); }
    __device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) const
    {
        return __shfl(v, src, 64);
    }
    __device__ __forceinline__ float shfl_f32(float v, int src) const { return __shfl(v, src, 64); }
    __device__ __forceinline__ int shfl_i32(int v, int src) const { return __shfl(v, src, 64); }
    __device__ __forceinline__ float shfl_up_f32(float v, int d) const { return __shfl_up(v, d, 64); }
    __device__ __forceinline__ float shfl_down_f32(float v, int d) const { return __shfl_down(v, d, 64); }
    __device__ __forceinline__ float shfl_xor_f32(float v, int m) const { return __shfl_xor(v, m, 64); }
    __device__ __forceinline__ int shfl_xor_i32(int v, int m) const { return __shfl_xor(v, m, 64); }
    __device__ __forceinline__ double shfl_xor_f64(double v, int m) const { return __shfl_xor(v, m, 64); }
    // value of `v` in lane `lane` (lane must be wave-uniform): v_readlane_b32, no LDS
    __device__ __forceinline__ int readlane_i32(int v, int lane) const { return __builtin_amdgcn_readlane(v, lane); }
    __device__ __forceinline__ int ctz64(unsigned long long v) const { return __ffsll((long long)v) - 1; }
    __device__ __forceinline__ void atomic_or64(unsigned long long* p, unsigned long long v) const { atomicOr(p, v); }
    __device__ __forceinline__ int atomic_add_i32(int* p, int v) const { return atomicAdd(p, v); }

    // ---- raw buffers and LDS-DMA (k_corr4d.h) ------------------------------------------------
    // A raw buffer: wave-uniform base + byte count; an access whose byte offset (the VGPR part)
    // lies beyond the count is dropped (stores) or returns zeros (loads) by the hardware.
    struct Buf {
        const void* base;
        unsigned nbytes;
    };
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    __device__ __forceinline__ int wave_id() const { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
    __device__ __forceinline__ Buf make_buf(const void* base, unsigned nbytes) const
    {
        // (descriptor inputs made provably wave-uniform: they must end up in scalar registers)
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)base);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)base >> 32));
        Buf b;
        b.base = (const void*)(((size_t)hi << 32) | (size_t)lo);
        b.nbytes = __builtin_amdgcn_readfirstlane(nbytes);
        return b;
    }
    // LDS byte address of a pointer into the workgroup's LDS (what the DMA takes as destination)
    __device__ __forceinline__ unsigned lds_addr(const void* p) const { return (unsigned)(size_t)p; }
    // Asynchronous global -> LDS copy, 16 bytes per lane: lane l of the wave writes LDS bytes
    // [lds_dst + 16 l, +16) with buffer bytes [byte_off, +16) (zeros where that is out of range).
    // `buffer_load_dwordx4 ... lds`: the data never passes through VGPRs, M0 carries the
    // wave-uniform destination.  The compiler does not see the transfer: it completes with
    // wait_dma() (vmcnt), and a barrier makes it visible to the other waves.
    // Cache policy of these streaming transfers (every byte is read once, apart from the windows' overlap):
    // AISX_DMA_POL = 0 none | 1 nt | 2 sc1 | 3 sc0 sc1 | 4 sc0 nt sc1 | 5 sc0 | 6 nt sc1 on the load, AISX_STORE_AUX = the aux bits of
    // buf_store64 (1 sc0, 2 nt, 16 sc1).  Defaults: what measured best (DESIGN_APPENDIX.md).
#ifndef AISX_DMA_POL
#define AISX_DMA_POL 1
#endif
#if AISX_DMA_POL == 0
#define AISX_DMA_POLICY ""
#elif AISX_DMA_POL == 1
#define AISX_DMA_POLICY " nt"
#elif AISX_DMA_POL == 2
#define AISX_DMA_POLICY " sc1"
#elif AISX_DMA_POL == 3
#define AISX_DMA_POLICY " sc0 sc1"
#elif AISX_DMA_POL == 4
#define AISX_DMA_POLICY " sc0 nt sc1"
#elif AISX_DMA_POL == 5
#define AISX_DMA_POLICY " sc0"
#elif AISX_DMA_POL == 6
#define AISX_DMA_POLICY " nt sc1"
#endif
#ifndef AISX_STORE_AUX
#define AISX_STORE_AUX 2
#endif
#ifndef AISX_LOAD_AUX // buf_load64 (k_corr4f.h's window loads): the same bits
#define AISX_LOAD_AUX AISX_STORE_AUX
#endif
    __device__ __forceinline__ void dma16(const Buf& b, unsigned byte_off, unsigned lds_dst) const
    {
        v4i w;
        w.x = (int)(unsigned)(size_t)b.base;
        w.y = (int)(unsigned)(((size_t)b.base >> 32) & 0xffffu);
        w.z = (int)b.nbytes;
        w.w = 0x00020000;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen" AISX_DMA_POLICY " lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(byte_off), "s"(w), "s"(dst)
                     : "memory");
    }
    // all of this wave's vector-memory operations (DMA included) have completed
    __device__ __forceinline__ void wait_dma() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    // ... all but the N youngest (they complete in order)
    template <int N>
    __device__ __forceinline__ void wait_vm() const
    {
        static_assert(N >= 0 && N < 64, "vmcnt");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    }
    // workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for
    // outstanding global stores or for DMA still in flight
    __device__ __forceinline__ void lds_barrier() const { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    // one complex item from buffer byte offset voff + soff (range-checked on voff only: zeros beyond the count), streaming
    // policy; the compiler keeps the count of loads in flight and waits where the value is first used
    __device__ __forceinline__ cf buf_load64(const Buf& b, unsigned voff, unsigned soff) const
    {
        const v2u d = __builtin_amdgcn_raw_buffer_load_b64(__builtin_amdgcn_make_buffer_rsrc((void*)b.base, 0, (int)b.nbytes, 0x00020000),
                                                           (int)voff, (int)soff, AISX_LOAD_AUX);
        return mk(__uint_as_float(d.x), __uint_as_float(d.y));
    }
    // one complex item to buffer byte offset voff + soff; range-checked on voff only (the scalar
    // part is added after the check)
    __device__ __forceinline__ void buf_store64(const Buf& b, unsigned voff, unsigned soff, cf v) const
    {
        v2u d;
        d.x = __float_as_uint(v.re);
        d.y = __float_as_uint(v.im);
        __builtin_amdgcn_raw_buffer_store_b64(d, __builtin_amdgcn_make_buffer_rsrc((void*)b.base, 0, (int)b.nbytes, 0x00020000),
                                              (int)voff, (int)soff, AISX_STORE_AUX);
    }
};

// The same context with the butterfly primitives left to the compiler (plain C++ complex
// arithmetic).  The F = 2048 correlator is bound by the latency of its L2 twiddle / spectrum
// reads at 3 waves per SIMD, not by VALU issue, and schedules better without opaque
// instructions in the way (measured: 1.20 ms against 1.37 ms per launch).
struct DevCtxC : DevCtx {
    __device__ __forceinline__ cf cadd(cf a, cf b) const { return mk(a.re + b.re, a.im + b.im); }
    __device__ __forceinline__ cf csub(cf a, cf b) const { return mk(a.re - b.re, a.im - b.im); }
    __device__ __forceinline__ cf add_mj(cf a, cf b) const { return mk(a.re + b.im, a.im - b.re); }
    __device__ __forceinline__ cf add_pj(cf a, cf b) const { return mk(a.re - b.im, a.im + b.re); }
    __device__ __forceinline__ cf cmul(cf a, cf b) const { return cmul_fma(a, b); }
    __device__ __forceinline__ cf cmul_conj(cf a, cf b) const { return cmul_conj_fma(a, b); }
    template <int KSEL, int CS, bool CNEG, int SS, bool SNEG>
    __device__ __forceinline__ cf cmul_sel(cf a) const
    {
        const float k0 = KSEL == 0 ? 0.92387953251128673848f : 0.70710678118654752440f;
        const float k1 = KSEL == 0 ? 0.38268343236508978178f : 0.70710678118654752440f;
        const float c = CNEG ? -(CS ? k1 : k0) : (CS ? k1 : k0), s = SNEG ? -(SS ? k1 : k0) : (SS ? k1 : k0);
        return mk(fmaf(-a.im, s, a.re * c), fmaf(a.re, s, a.im * c));
    }
};

} // namespace aisx
