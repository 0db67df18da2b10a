// cu_mask.hip -- where do the workgroups of a stream made with hipExtStreamCreateWithCUMask land on gfx950?
// A kernel of 2048 one-wave workgroups (80 KiB of LDS each: at most one... two per CU) records XCC_ID and
// HW_ID; the host counts distinct (xcc, se, sh, cu) for masks of the first N bits and times a
// latency-bound kernel (a dependent FMA chain, one wave per workgroup, 128 workgroups x 92 KiB LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

__global__ void k_where(unsigned* out)
{
    extern __shared__ char smem[];
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
    }
    // stay a while, so that the workgroups spread
    float a = (float)threadIdx.x;
    for (int i = 0; i < 20000; i++)
        a = __builtin_fmaf(a, 1.0001f, 0.5f);
    if (a == 12345.f)
        out[0] = 0;
    (void)smem;
}

__global__ void k_chain(float* out, int iters)
{
    extern __shared__ char smem[];
    float a = (float)threadIdx.x;
    for (int i = 0; i < iters; i++)
        a = __builtin_fmaf(a, 1.0001f, 0.5f);
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    (void)smem;
}

#define CK(e)                                                                   \
    do {                                                                        \
        hipError_t e__ = (e);                                                   \
        if (e__ != hipSuccess) {                                                \
            printf("%s: %s\n", #e, hipGetErrorString(e__));                     \
            return 1;                                                           \
        }                                                                       \
    } while (0)

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("%s: %d CUs\n", prop.gcnArchName, ncu);
    unsigned* d;
    float* f;
    const int NB = 2048;
    CK(hipMalloc(&d, NB * 8));
    CK(hipMalloc(&f, 1024 * 256 * 4));
    CK(hipFuncSetAttribute((const void*)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CK(hipFuncSetAttribute((const void*)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 92 * 1024));
    const int ns[] = { 0, 8, 32, 48, 64, 96, 128, 192, 256 };
    for (int n : ns) {
        hipStream_t s;
        if (n == 0) {
            CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        } else {
            std::vector<uint32_t> m((size_t)(ncu + 31) / 32, 0u);
            for (int b = 0; b < n; b++)
                m[(size_t)b / 32] |= 1u << (b % 32);
            CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data()));
        }
        CK(hipMemsetAsync(d, 0xff, NB * 8, s));
        hipLaunchKernelGGL(k_where, dim3(NB), dim3(64), 80 * 1024, s, d);
        CK(hipStreamSynchronize(s));
        std::vector<unsigned> h(2 * NB);
        CK(hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost));
        std::set<unsigned> cus;
        int per_xcc[16] = { 0 };
        std::set<unsigned> cx[16];
        for (int b = 0; b < NB; b++) {
            const unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const unsigned id = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            cus.insert(id);
            cx[xcc].insert(id);
        }
        for (int x = 0; x < 16; x++)
            per_xcc[x] = (int)cx[x].size();
        // the latency-bound kernel: 128 workgroups of 4 waves, 92 KiB LDS (one per CU)
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_chain, dim3(128), dim3(256), 92 * 1024, s, f, 100000);
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(k_chain, dim3(128), dim3(256), 92 * 1024, s, f, 400000);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("mask bits [0,%3d): %3zu distinct CUs; per XCC:", n, cus.size());
        for (int x = 0; x < 8; x++)
            printf(" %2d", per_xcc[x]);
        printf("   128 WGs x 92 KiB, dependent chain: %.3f ms\n", ms);
        CK(hipStreamDestroy(s));
    }
    return 0;
}
