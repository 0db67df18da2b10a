// micro-benchmark: cost of one instruction of each kind for a lone wave (dependent chains,
// 64 per loop trip), design input for the timing-recovery kernel
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

#define KERNEL(name, decl, body)                                   \
    __global__ void name(float* out, int iters)                    \
    {                                                              \
        decl;                                                      \
        for (int i = 0; i < iters; i++) {                          \
            asm volatile(REP64(body) : "+v"(a), "+v"(b), "+v"(c) : "v"(d) : "vcc", "scc", "s20"); \
        }                                                          \
        out[threadIdx.x] = a.x + b.x + c.x;                        \
    }

typedef float f2 __attribute__((ext_vector_type(2)));
#define DECL f2 a = { threadIdx.x * 1e-3f, 1.f }, b = { 1.0001f, 0.9999f }, c = { 0.5f, 0.25f }, d = { 1.00001f, 1.f }

KERNEL(k_pk_fma, DECL, "v_pk_fma_f32 %0, %0, %3, %2\n")
KERNEL(k_pk_mul, DECL, "v_pk_mul_f32 %0, %0, %3\n")
KERNEL(k_pk_add, DECL, "v_pk_add_f32 %0, %0, %3\n")
KERNEL(k_pk_mul_opsel, DECL, "v_pk_mul_f32 %0, %0, %3 op_sel:[0,1] op_sel_hi:[1,0]\n")
KERNEL(k_pk_muladd, DECL, "v_pk_mul_f32 %1, %0, %3\n v_pk_add_f32 %0, %1, %2\n")
KERNEL(k_pk_muladd_indep, DECL, "v_pk_mul_f32 %1, %2, %3\n v_pk_add_f32 %0, %0, %3\n")

typedef float f1;
#define KERNEL1(name, body)                                        \
    __global__ void name(float* out, int iters)                    \
    {                                                              \
        float a = threadIdx.x * 1e-3f + 1.f, b = 1.0001f, c = 0.5f; \
        int ia = threadIdx.x, ib = 3;                              \
        for (int i = 0; i < iters; i++) {                          \
            asm volatile(REP64(body) : "+v"(a), "+v"(b), "+v"(ia) : "v"(c), "v"(ib) : "vcc", "scc", "s20"); \
        }                                                          \
        out[threadIdx.x] = a + b + ia;                             \
    }
KERNEL1(k_add, "v_add_f32 %0, %0, %3\n")
KERNEL1(k_mul, "v_mul_f32 %0, %0, %3\n")
KERNEL1(k_floor, "v_floor_f32 %0, %0\n")
KERNEL1(k_rndne, "v_rndne_f32 %0, %0\n")
KERNEL1(k_cvt, "v_cvt_i32_f32 %2, %0\n v_cvt_f32_i32 %0, %2\n")
KERNEL1(k_mad24, "v_mad_u32_u24 %2, %2, 3, %4\n")
KERNEL1(k_lshladd, "v_lshl_add_u32 %2, %2, 1, %4\n")
KERNEL1(k_and, "v_and_b32 %2, %2, %4\n")
KERNEL1(k_add_indep2, "v_add_f32 %0, %0, %3\n v_add_f32 %1, %1, %3\n")
KERNEL1(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %3\n v_cndmask_b32 %0, %0, %1, vcc\n")
KERNEL1(k_cmp_salu, "v_cmp_lt_f32 vcc, %0, %3\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %0, %0, %1, vcc\n")
KERNEL1(k_nop_add, "s_nop 0\n v_add_f32 %0, %0, %3\n")
KERNEL1(k_salu_chain, "s_add_u32 s20, s20, 1\n")
KERNEL1(k_valu_salu_mix, "v_add_f32 %0, %0, %3\n s_add_u32 s20, s20, 1\n")

template <class K>
static void run(const char* name, K k, float* d, int per_trip)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k<<<1, 64>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-22s %.2f ns per instruction (%d per group)\n", name, ms * 1e6 / (iters * 64.0 * per_trip), per_trip);
    fflush(stdout);
}

int main()
{
    float* d;
    (void)hipMalloc(&d, 1 << 16);
    run("v_pk_fma_f32 dep", k_pk_fma, d, 1);
    run("v_pk_mul_f32 dep", k_pk_mul, d, 1);
    run("v_pk_add_f32 dep", k_pk_add, d, 1);
    run("v_pk_mul op_sel dep", k_pk_mul_opsel, d, 1);
    run("pk_mul->pk_add dep", k_pk_muladd, d, 2);
    run("pk_mul,pk_add indep", k_pk_muladd_indep, d, 2);
    run("v_add_f32 dep", k_add, d, 1);
    run("v_mul_f32 dep", k_mul, d, 1);
    run("v_floor_f32 dep", k_floor, d, 1);
    run("v_rndne_f32 dep", k_rndne, d, 1);
    run("cvt i32<->f32 dep", k_cvt, d, 2);
    run("v_mad_u32_u24 dep", k_mad24, d, 1);
    run("v_lshl_add_u32 dep", k_lshladd, d, 1);
    run("v_and_b32 dep", k_and, d, 1);
    run("v_add_f32 x2 indep", k_add_indep2, d, 2);
    run("v_cmp->v_cndmask", k_cmp_cnd, d, 2);
    run("v_cmp->s_and->cndmask", k_cmp_salu, d, 3);
    run("s_nop + v_add", k_nop_add, d, 2);
    run("s_add_u32 dep", k_salu_chain, d, 1);
    run("v_add + s_add mix", k_valu_salu_mix, d, 2);
    return 0;
}
