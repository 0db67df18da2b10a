// micro-benchmark: cost of a burst of LDS reads issued by a lone wave, as the
// timing-recovery pair loop does (4 x ds_read2st64_b64 + 2 x ds_read_b128 per FIR)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NB64, int NB128, int NVALU>
__global__ void burst(float* out, int iters)
{
    extern __shared__ char smem[];
    float* f = (float*)smem;
    for (int i = threadIdx.x; i < 32768; i += 64)
        f[i] = 1e-3f * i;
    __syncthreads();
    unsigned a = threadIdx.x * 8;          // slot-major sample address
    unsigned t = 135168 % 100000 + (threadIdx.x % 37) * 48; // a tap row
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        float4 s[8];
        float4 w[4];
#pragma unroll
        for (int k = 0; k < NB64; k++)
            asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(s[k]) : "v"(a), "n"(2 * k), "n"(2 * k + 1));
#pragma unroll
        for (int k = 0; k < NB128; k++)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[k]) : "v"(t), "n"(16 * k));
        asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int k = 0; k < NB64; k++)
            acc += s[k].x;
#pragma unroll
        for (int k = 0; k < NB128; k++)
            acc += w[k].x;
#pragma unroll
        for (int k = 0; k < NVALU; k++)
            asm volatile("v_add_f32 %0, %0, %0" : "+v"(acc));
        // the next addresses depend on the result, as in the recurrence
        a = (a + ((unsigned)(acc != 12345.f) << 9)) & 0x1ffff;
    }
    out[threadIdx.x] = acc;
}

template <class K>
static void run(const char* name, K k, float* d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    float ms = 0;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k<<<1, 64, 140000>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-34s %.1f ns per burst\n", name, ms * 1e6 / iters);
    fflush(stdout);
}

int main()
{
    float* d;
    (void)hipMalloc(&d, 1 << 16);
    run("0 loads, 40 valu", burst<0, 0, 40>, d);
    run("1 x b64x2", burst<1, 0, 40>, d);
    run("4 x b64x2 (one FIR samples)", burst<4, 0, 40>, d);
    run("4 x b64x2 + 2 x b128 (one FIR)", burst<4, 2, 40>, d);
    run("8 x b64x2 + 4 x b128 (pair)", burst<8, 4, 40>, d);
    run("8 x b64x2", burst<8, 0, 40>, d);
    run("4 x b128", burst<0, 4, 40>, d);
    return 0;
}
