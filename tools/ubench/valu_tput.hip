// valu_tput.hip -- VALU issue throughput on MI355X with 1..4 waves per SIMD: cycles per wave64
// instruction for plain and packed fp32 (what bounds the FFT correlator's butterflies).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
#define R8(x, a) x(a##0) x(a##1) x(a##2) x(a##3) x(a##4) x(a##5) x(a##6) x(a##7)

#define KERN(name, INSTR)                                                                             \
    __global__ __launch_bounds__(256) void name(float* out, int iters)                               \
    {                                                                                                 \
        f2 a0 = { threadIdx.x * 1e-3f, 1.f }, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f,  \
           a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;                                               \
        f2 d = { 1.00001f, 0.99999f };                                                                \
        for (int i = 0; i < iters; i++) {                                                             \
            asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)      \
                         INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)      \
                         INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)      \
                         INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)      \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(d));                                                                   \
        }                                                                                             \
        f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                 \
        if (s.x == 123.456f) out[threadIdx.x] = s.x + s.y;                                            \
    }
#define KERN1(name, INSTR)                                                                             \
    __global__ __launch_bounds__(256) void name(float* out, int iters)                               \
    {                                                                                                 \
        float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f,  \
           a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;                                               \
        float d = 1.00001f;                                                                \
        for (int i = 0; i < iters; i++) {                                                             \
            asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)      \
                         INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)      \
                         INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)      \
                         INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)      \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(d));                                                                   \
        }                                                                                             \
        float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                 \
        if (s == 123.456f) out[threadIdx.x] = s;                                            \
    }
#define I_ADD(k) "v_add_f32 %" #k ", %" #k ", %8\n"
#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %8\n"
#define I_PKADD(k) "v_pk_add_f32 %" #k ", %" #k ", %8\n"
#define I_PKADDSEL(k) "v_pk_add_f32 %" #k ", %" #k ", %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
#define I_PKMUL(k) "v_pk_mul_f32 %" #k ", %" #k ", %8\n"
#define I_PKFMA(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %8\n"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n"
#define I_PKMOV(k) "v_pk_mov_b32 %" #k ", %" #k ", %8\n"
KERN1(k_add, I_ADD)
KERN1(k_fma, I_FMA)
KERN(k_pkadd, I_PKADD)
KERN(k_pkaddsel, I_PKADDSEL)
KERN(k_pkmul, I_PKMUL)
KERN(k_pkfma, I_PKFMA)
KERN1(k_mov, I_MOV)
KERN(k_pkmov, I_PKMOV)

template <class K>
static void run(const char* name, K k, float* d)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 20000;
    printf("%-12s", name);
    for (int wps = 1; wps <= 4; wps++) {
        // wps workgroups of 256 threads per CU: wps waves on every SIMD
        hipLaunchKernelGGL(k, dim3(256 * wps), dim3(256), 0, 0, d, 100);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256 * wps), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 32 * wps;
        printf("  %dw/simd: %.2f ns/instr/simd", wps, ms * 1e6 / instr_per_simd);
    }
    printf("\n");
}

int main()
{
    float* d;
    (void)hipMalloc(&d, 4096);
    run("v_add", k_add, d);
    run("v_fma", k_fma, d);
    run("v_pk_add", k_pkadd, d);
    run("v_pk_add_sel", k_pkaddsel, d);
    run("v_pk_mul", k_pkmul, d);
    run("v_pk_fma", k_pkfma, d);
    run("v_mov", k_mov, d);
    run("v_pk_mov", k_pkmov, d);
    return 0;
}
