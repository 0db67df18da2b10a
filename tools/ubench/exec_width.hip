// micro-benchmark: does a wave64 VALU instruction get cheaper when only 16 or 32 lanes are
// active?  (design input for the timing-recovery kernel: channels per wave)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(float* out, int active, int iters, int indep)
{
    const int l = threadIdx.x;
    if (l >= active)
        return;
    float a = (float)l * 1e-3f, b = 1.0001f, c = 0.5f;
    float a2 = a + 1.f, a3 = a + 2.f, a4 = a + 3.f;
    if (indep) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                a = __builtin_fmaf(a, b, c);
                a2 = __builtin_fmaf(a2, b, c);
                a3 = __builtin_fmaf(a3, b, c);
                a4 = __builtin_fmaf(a4, b, c);
            }
        }
    } else {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 64; k++)
                a = __builtin_fmaf(a, b, c);
        }
    }
    out[blockIdx.x * 64 + l] = a + a2 + a3 + a4;
}
__global__ void lds_chase(int* out, int iters)
{
    __shared__ int nxt[1024];
    for (int i = threadIdx.x; i < 1024; i += 64)
        nxt[i] = (i + 64) & 1023;
    __syncthreads();
    int p = threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++)
            p = nxt[p];
    }
    out[threadIdx.x] = p;
}
// a taken branch per step: the loop is not unrolled and carries one dependent fma
__global__ void branchy(float* out, int iters)
{
    float a = threadIdx.x * 1e-3f;
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        a = __builtin_fmaf(a, 1.0001f, 0.5f);
        asm volatile("" ::: "memory");
    }
    out[threadIdx.x] = a;
}
int main()
{
    float* d;
    hipMalloc(&d, 1 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int indep = 0; indep < 2; indep++)
        for (int active : { 64, 32, 16, 1 })
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                chain<<<1, 64>>>(d, active, iters, indep);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep)
                    printf("indep=%d active=%2d: %.3f ms, %.2f ns/instr\n", indep, active, ms,
                           ms * 1e6 / (iters * 64.0));
            }
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        lds_chase<<<1, 64>>>((int*)d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep)
            printf("lds pointer chase: %.2f ns per dependent ds_read\n", ms * 1e6 / (iters * 16.0));
    }
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        branchy<<<1, 64>>>(d, iters * 16);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep)
            printf("loop step (fma + s_add + s_cmp + taken branch): %.2f ns\n", ms * 1e6 / (iters * 16.0));
    }
    return 0;
}
