// micro-benchmark: does a burst of LDS reads get cheaper for a lone wave when only 16 of its 64
// lanes are enabled?  (the timing-recovery pair loop needs 2 lanes per channel, 8 channels per
// wave: lanes 0-7 and 16-23; the other 48 repeat their work)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NB64, int NB128, int NVALU, int LANES>
__global__ void burst(float* out, int iters)
{
    extern __shared__ char smem[];
    float* f = (float*)smem;
    for (int i = threadIdx.x; i < 32768; i += blockDim.x)
        f[i] = 1e-3f * i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned a = (lane & 7) * 8 + (threadIdx.x >> 6) * 16896;          // slot-major sample address, one ring set per wave
    unsigned t = 135168 % 100000 + (lane % 37) * 48; // a tap row
    float acc = 0.f;
    const bool on = LANES == 64 ? true : (LANES == 32 ? (lane & 0x20) == 0 : (LANES == 16 ? (lane & 0x28) == 0 : (lane & 0x38) == 0));
    if (on) {
        for (int i = 0; i < iters; i++) {
            float4 s[8];
            float4 w[4];
#pragma unroll
            for (int k = 0; k < NB64; k++)
                asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(s[k]) : "v"(a), "n"(16 * k), "n"(16 * k + 8));
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
            a = (a + ((unsigned)(acc != 12345.f) << 7)) & 0x3fff;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <class K>
static void run(const char* name, K k, float* d, int wgs, int threads)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    float ms = 0;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k<<<wgs, threads, 140000>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-44s wgs %3d x %3d threads: %.1f ns per burst\n", name, wgs, threads, ms * 1e6 / iters);
    fflush(stdout);
}

int main()
{
    float* d;
    (void)hipMalloc(&d, 1 << 20);
    for (int cfg = 0; cfg < 2; cfg++) {
        const int wgs = cfg ? 128 : 1, th = cfg ? 256 : 64;
        run("0 loads, 40 valu, 64 lanes", burst<0, 0, 40, 64>, d, wgs, th);
        run("0 loads, 40 valu, 16 lanes", burst<0, 0, 40, 16>, d, wgs, th);
        run("FIR (4 x b64x2 + 2 x b128), 64 lanes", burst<4, 2, 40, 64>, d, wgs, th);
        run("FIR, 32 lanes", burst<4, 2, 40, 32>, d, wgs, th);
        run("FIR, 16 lanes (0-7, 16-23)", burst<4, 2, 40, 16>, d, wgs, th);
        run("FIR, 8 lanes", burst<4, 2, 40, 8>, d, wgs, th);
        run("samples only 4 x b64x2, 64 lanes", burst<4, 0, 40, 64>, d, wgs, th);
        run("samples only, 16 lanes", burst<4, 0, 40, 16>, d, wgs, th);
        run("taps only 2 x b128, 64 lanes", burst<0, 2, 40, 64>, d, wgs, th);
        run("taps only, 16 lanes", burst<0, 2, 40, 16>, d, wgs, th);
    }
    return 0;
}
