// micro-benchmark (round 6): do a CU's LDS traffic and its VALU work overlap when every wave runs the
// correlator's per-pass pattern -- R x ds_read_b64, wait, P packed-fp32 instructions, W x ds_write_b64 --
// at the correlator's occupancy (512-thread workgroups, two per CU, four waves per SIMD)?
// Prints cycles per pass and CU (clock64 of one wave) for: LDS only, VALU only, both, and both with
// the stores spread between the arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

// MODE 0: reads + writes only; 1: VALU only; 2: reads, VALU, writes; 3: reads, VALU with a write after every P / W ops
// the LDS-only pattern with D buffer_load_dwordx4 ... lds (1 KiB each, streaming from `src`) per pass and wave, waited for
// a pass later: what the LDS-DMA's writes cost the LDS beside the waves' own traffic
typedef int v4i __attribute__((ext_vector_type(4)));
template <int R, int W, int DNUM, int DDEN, int P>
__global__ __launch_bounds__(512, 4) void dma_loop(float* out, int iters, const float* src, unsigned nbytes)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned t = threadIdx.x;
    unsigned a = (t >> 6) * (64 * 8 * 8) + (t & 63) * 8;
    v2f x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        x[k].x = 1e-3f * (float)(t + k);
        x[k].y = 1.f;
        *(v2f*)(smem + a + k * 512) = x[k];
    }
    v2f c;
    c.x = 0.999f;
    c.y = 1e-4f;
    v4i rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)src);
    rs.y = __builtin_amdgcn_readfirstlane((int)(((size_t)src >> 32) & 0xffff));
    rs.z = __builtin_amdgcn_readfirstlane((int)nbytes);
    rs.w = 0x00020000;
    const unsigned wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned dst0 = (unsigned)(size_t)smem + 32768u + wave * 4096u; // (behind the waves' own 32 KiB)
    unsigned off = ((blockIdx.x * 8 + wave) * 1024u * 256u + (t & 63) * 16u) % (nbytes - 1024u * 1024u);
    __syncthreads();
    for (int i = 0; i < iters; i++) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); // (six transfers in flight: the HBM latency is not what is measured)
        if (DNUM > 0 && (i % DDEN) < DNUM) {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen nt lds" : : "v"(off), "s"(rs), "s"(dst0 + (unsigned)(i & 3) * 1024u) : "memory", "m0");
            off += 1024u;
        }
#pragma unroll
        for (int k = 0; k < R; k++)
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[k & 7]) : "v"(a), "n"((k & 7) * 512));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < P; k++)
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[k & 7]) : "v"(c), "v"(x[(k + 3) & 7]));
#pragma unroll
        for (int k = 0; k < W; k++)
            asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(x[k & 7]), "n"((k & 7) * 512) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
        s += x[k].x + x[k].y;
    out[blockIdx.x * 512 + t] = s;
}

template <class K>
static void run_dma(const char* name, K k, float* d, const float* src, unsigned nbytes)
{
    const int iters = 4000, lds = 73728;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        k<<<512, 512, lds>>>(d, iters, src, nbytes);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-58s %7.1f ns per pass (16 waves per CU)\n", name, ms * 1e6 / iters);
    fflush(stdout);
}

template <int MODE, int R, int W, int P, bool B128>
__global__ __launch_bounds__(512, 4) void pass_loop(float* out, int iters, long long* cyc)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned t = threadIdx.x;
    // a wave-private region of 64 x 8 slots, lane-contiguous (conflict free)
    unsigned a = (t >> 6) * (64 * 8 * 8) + (t & 63) * 8;
    v2f x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        x[k].x = 1e-3f * (float)(t + k);
        x[k].y = 1.f;
        *(v2f*)(smem + a + k * 512) = x[k];
    }
    v2f c;
    c.x = 0.999f;
    c.y = 1e-4f;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (MODE != 1) {
#pragma unroll
            for (int k = 0; k < R; k++)
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[k & 7]) : "v"(a), "n"((k & 7) * 512));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < W; k++)
                asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(x[k & 7]), "n"((k & 7) * 512) : "memory");
        } else if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int k = 0; k < P; k++)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[k & 7]) : "v"(c), "v"(x[(k + 3) & 7]));
            if (MODE == 2) {
#pragma unroll
                for (int k = 0; k < W; k++)
                    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(x[k & 7]), "n"((k & 7) * 512) : "memory");
            }
        } else {
#pragma unroll
            for (int k = 0; k < P; k++) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[k & 7]) : "v"(c), "v"(x[(k + 3) & 7]));
                if (W > 0 && (k + 1) % (P / W) == 0 && (k + 1) / (P / W) <= W)
                    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(x[((k + 1) / (P / W) - 1) & 7]), "n"((((k + 1) / (P / W) - 1) & 7) * 512) : "memory");
            }
        }
    }
    const long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
        s += x[k].x + x[k].y;
    out[blockIdx.x * 512 + t] = s;
    if (t == 0 && blockIdx.x == 0)
        *cyc = t1 - t0;
}

template <class K>
static void run(const char* name, K k, float* d, long long* dc, int wg_per_cu)
{
    const int iters = 4000, lds = wg_per_cu == 2 ? 73728 : 36864;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    long long c = 0;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        k<<<256 * wg_per_cu, 512, lds>>>(d, iters, dc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    // clock64 counts at a constant 100 MHz on this part: report wall time per pass (all waves of a CU run the loop side by side)
    printf("%-58s %7.1f ns per pass and wave-set (%d waves per CU)   [s_memtime ticks per pass %.2f]\n", name, ms * 1e6 / iters,
           wg_per_cu * 8, (double)c / iters);
    fflush(stdout);
}

int main()
{
    float* d;
    long long* dc;
    (void)hipMalloc(&d, 4 << 20);
    (void)hipMalloc(&dc, 8);
    // spin the clocks up
    for (int i = 0; i < 30; i++)
        pass_loop<1, 8, 8, 42, false><<<512, 512, 36864>>>(d, 4000, dc);
    (void)hipDeviceSynchronize();
    for (int wg = 1; wg <= 2; wg++) {
        printf("--- %d workgroup(s) of 512 threads per CU\n", wg);
        run("LDS only: 8 x ds_read_b64 + 8 x ds_write_b64", pass_loop<0, 8, 8, 42, false>, d, dc, wg);
        run("LDS only: 8 x ds_read_b64", pass_loop<0, 8, 0, 42, false>, d, dc, wg);
        run("LDS only: 8 x ds_write_b64 (after 1 read)", pass_loop<0, 1, 8, 42, false>, d, dc, wg);
        run("VALU only: 42 x v_pk_fma_f32", pass_loop<1, 8, 8, 42, false>, d, dc, wg);
        run("8 reads, 42 pk, 8 writes", pass_loop<2, 8, 8, 42, false>, d, dc, wg);
        run("8 reads, 42 pk with a write after every 5th", pass_loop<3, 8, 8, 42, false>, d, dc, wg);
        run("VALU only: 84 x v_pk_fma_f32", pass_loop<1, 8, 8, 84, false>, d, dc, wg);
        run("8 reads, 84 pk, 8 writes", pass_loop<2, 8, 8, 84, false>, d, dc, wg);
        run("8 reads, 84 pk with a write after every 10th", pass_loop<3, 8, 8, 84, false>, d, dc, wg);
    }
    float* src;
    const unsigned nbytes = 3u << 30;
    (void)hipMalloc(&src, nbytes);
    (void)hipMemset(src, 0, nbytes);
    printf("--- LDS-DMA beside the waves' own LDS traffic (2 workgroups per CU)\n");
    run_dma("8 r + 8 w, no DMA", dma_loop<8, 8, 0, 1, 0>, d, src, nbytes);
    run_dma("8 r + 8 w, 1 DMA (1 KiB) per wave every 4th pass", dma_loop<8, 8, 1, 4, 0>, d, src, nbytes);
    run_dma("8 r + 8 w, 1 DMA per wave every 2nd pass", dma_loop<8, 8, 1, 2, 0>, d, src, nbytes);
    run_dma("8 r + 8 w, 1 DMA per wave and pass", dma_loop<8, 8, 1, 1, 0>, d, src, nbytes);
    run_dma("8 r + 42 pk + 8 w, no DMA", dma_loop<8, 8, 0, 1, 42>, d, src, nbytes);
    run_dma("8 r + 42 pk + 8 w, 1 DMA per wave every 2nd pass", dma_loop<8, 8, 1, 2, 42>, d, src, nbytes);
    run_dma("8 r + 42 pk + 8 w, 1 DMA per wave and pass", dma_loop<8, 8, 1, 1, 42>, d, src, nbytes);
    run_dma("42 pk only, 1 DMA per wave and pass", dma_loop<0, 0, 1, 1, 42>, d, src, nbytes);
    run_dma("42 pk only, no DMA", dma_loop<0, 0, 0, 1, 42>, d, src, nbytes);
    return 0;
}
