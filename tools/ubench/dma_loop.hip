// dma_loop.hip -- the refill pattern of the time-parallel timing recovery in isolation: per round 8 LDS-DMA
// instructions (16 streams x 64 B, uncached), a burst of LDS reads and some arithmetic, and a wait that
// leaves the two youngest rounds in flight.  Prints cycles per round spent issuing / waiting / computing.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
template <bool READS>
__global__ void k_loop(const float* src, unsigned nbytes, long long* out, float* sink, int rounds)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lane = threadIdx.x & 63;
    v4i rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)src);
    rs.y = __builtin_amdgcn_readfirstlane((int)(((size_t)src >> 32) & 0xffff));
    rs.z = __builtin_amdgcn_readfirstlane((int)nbytes);
    rs.w = 0x00020000;
    const unsigned base = ((blockIdx.x & 1023) * 16u + (lane & 15)) * 65536u + (lane >> 4) * 16u;
    long long t_issue = 0, t_wait = 0, t_comp = 0;
    float acc = 0.f;
    for (int r = 0; r < rounds; r++) {
        const long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem + (unsigned)(((r & 3) * 8 + k)) * 1024u);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(base + (unsigned)(r * 8 + k) * 64u), "s"(rs), "s"(dst) : "memory", "m0");
        }
        const long long t1 = __builtin_readcyclecounter();
        if (READS) {
            const float* l = (const float*)smem;
#pragma unroll 16
            for (int k = 0; k < 64; k++)
                acc += l[((k * 67 + lane) & 8191)] * 1.0001f;
        } else {
#pragma unroll 16
            for (int k = 0; k < 256; k++)
                acc = acc * 1.0001f + 0.5f;
        }
        const long long t2 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        const long long t3 = __builtin_readcyclecounter();
        t_issue += t1 - t0;
        t_comp += t2 - t1;
        t_wait += t3 - t2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && blockIdx.x == 0) {
        out[0] = t_issue / rounds;
        out[1] = t_comp / rounds;
        out[2] = t_wait / rounds;
    }
    if (acc == 12345.f)
        sink[0] = acc;
}
int main()
{
    float* d; long long* o; float* sink;
    (void)hipMalloc(&d, (size_t)1 << 30);
    (void)hipMalloc(&o, 64);
    (void)hipMalloc(&sink, 64);
    for (int nwg : { 1, 256, 1024 }) {
        for (int reads = 0; reads < 2; reads++) {
            if (reads) {
                (void)hipFuncSetAttribute((const void*)k_loop<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
                hipLaunchKernelGGL((k_loop<true>), dim3(nwg), dim3(64), 32768, 0, d, 1u << 30, o, sink, 64);
            } else {
                (void)hipFuncSetAttribute((const void*)k_loop<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
                hipLaunchKernelGGL((k_loop<false>), dim3(nwg), dim3(64), 32768, 0, d, 1u << 30, o, sink, 64);
            }
            (void)hipDeviceSynchronize();
            long long h[3];
            (void)hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
            printf("%4d waves, %s between: per round issue %lld, compute %lld, wait %lld cycles\n", nwg, reads ? "LDS reads" : "arithmetic", h[0], h[1], h[2]);
        }
    }
    return 0;
}
