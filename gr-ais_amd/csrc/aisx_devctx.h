// aisx_devctx.h -- the gfx950 execution context the kernel bodies run under in
// the product: wave64 ballot/shuffle builtins, LDS, workgroup barrier, global
// atomics.  (tests/emul has the CPU model of the same interface.)
#pragma once
#include <hip/hip_runtime.h>

namespace aisx {

struct DevCtx {
    char* lds_;
    __device__ __forceinline__ int tid() const { return threadIdx.x; }
    __device__ __forceinline__ int nthreads() const { return blockDim.x; }
    __device__ __forceinline__ int bx() const { return blockIdx.x; }
    __device__ __forceinline__ int by() const { return blockIdx.y; }
    __device__ __forceinline__ char* lds() const { return lds_; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    // a lane that leaves the kernel for good before its workgroup's barriers (nothing to do
    // on the device: the hardware counts waves, not lanes)
    __device__ __forceinline__ void retire() const {}
    __device__ __forceinline__ unsigned long long ballot(bool p) const { return __builtin_amdgcn_ballot_w64(p); }
    // per-lane predicate from a wave-uniform mask (the mask goes straight into exec / vcc)
    __device__ __forceinline__ bool inv_ballot(unsigned long long m) const { return __builtin_amdgcn_inverse_ballot_w64(m); }
    // x - floorf(x) for x >= 0 (v_fract_f32 is exact there)
    __device__ __forceinline__ float fract(float x) const { return __builtin_amdgcn_fractf(x); }
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
};

} // namespace aisx
