// dma_issue.hip -- what issuing vector-memory instructions costs ONE wave on gfx950 when the lanes'
// addresses are spread out: LDS-DMA (`buffer_load_dwordx4 ... lds`) against plain buffer loads into
// registers, for contiguous, per-stream (16 streams x 64 B) and per-lane (64 lines) address patterns.
// Prints shader cycles per instruction until the last one has ISSUED and until all have completed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int PAT, bool LDS, int K>
__global__ void k_issue(const float* src, unsigned nbytes, long long* out, float* sink)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lane = threadIdx.x & 63;
    v4i rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)src);
    rs.y = __builtin_amdgcn_readfirstlane((int)(((size_t)src >> 32) & 0xffff));
    rs.z = __builtin_amdgcn_readfirstlane((int)nbytes);
    rs.w = 0x00020000;
    // byte offset of this lane for instruction k
    auto addr = [&](int k) -> unsigned {
        if (PAT == 0) return (unsigned)k * 1024u + lane * 16u;                                   // one contiguous KiB
        if (PAT == 1) return lane * 65536u + (unsigned)k * 16u + (blockIdx.x & 7) * 4096u;        // 64 streams, 16 B each
        if (PAT == 3) // 16 streams x 64 B, every block its own memory (nothing cached): stream s of block b at (b * 16 + s) * 64 KiB
            return ((blockIdx.x & 1023) * 16u + (lane & 15)) * 65536u + (unsigned)k * 64u + (lane >> 4) * 16u;
        return (lane & 15) * 262144u + (unsigned)k * 64u + (lane >> 4) * 16u + (blockIdx.x & 7) * 8192u; // 16 streams x 64 B
    };
    v4f acc = { 0, 0, 0, 0 };
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t0 = __builtin_readcyclecounter();
    if (LDS) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem + (unsigned)(k & 15) * 1024u);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(addr(k)), "s"(rs), "s"(dst) : "memory", "m0");
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; k++) {
            v4f v;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(addr(k)), "s"(rs) : "memory");
            acc += v; // (the compiler cannot know the load is pending: the sum is garbage, the timing is not)
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = t2 - t0;
    }
    if (acc.x == 12345.f)
        sink[0] = acc.x;
}

template <int PAT, bool LDS, int K>
static void run(const char* name, const float* d, unsigned nbytes, long long* o, float* sink, int nwg)
{
    (void)hipFuncSetAttribute((const void*)k_issue<PAT, LDS, K>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384);
    long long best[2] = { 1LL << 60, 1LL << 60 };
    for (int rep = 0; rep < (PAT == 3 ? 1 : 5); rep++) {
        hipLaunchKernelGGL((k_issue<PAT, LDS, K>), dim3(nwg), dim3(64), 16384, 0, d, nbytes, o, sink);
        (void)hipDeviceSynchronize();
        long long h[2];
        (void)hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
        for (int i = 0; i < 2; i++)
            best[i] = h[i] < best[i] ? h[i] : best[i];
    }
    printf("%-44s %2d waves/CU-ish (%4d wg): issued after %6.0f cycles/instr, complete after %6.0f cycles/instr\n", name, nwg / 256, nwg,
           (double)best[0] / K, (double)best[1] / K);
}

int main()
{
    const size_t N = (size_t)1 << 30; // 256 MiB of floats = enough for 64 streams 64 KiB apart ... use bytes
    float* d;
    long long* o;
    float* sink;
    (void)hipMalloc(&d, N);
    (void)hipMemset(d, 0, N);
    (void)hipMalloc(&o, 64);
    (void)hipMalloc(&sink, 64);
    const unsigned nb = (unsigned)N;
    for (int nwg : { 1, 1024 }) {
        run<0, true, 32>("LDS-DMA, contiguous KiB", d, nb, o, sink, nwg);
        run<2, true, 32>("LDS-DMA, 16 streams x 64 B", d, nb, o, sink, nwg);
        run<1, true, 32>("LDS-DMA, 64 streams x 16 B", d, nb, o, sink, nwg);
        run<0, false, 32>("register load, contiguous KiB", d, nb, o, sink, nwg);
        run<2, false, 32>("register load, 16 streams x 64 B", d, nb, o, sink, nwg);
        run<1, false, 32>("register load, 64 streams x 16 B", d, nb, o, sink, nwg);
        run<3, true, 64>("LDS-DMA, 16 streams x 64 B, uncached", d, nb, o, sink, nwg);
        run<3, false, 64>("register load, 16 streams x 64 B, uncached", d, nb, o, sink, nwg);
    }
    return 0;
}
