// traffic_calib.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access
// shapes the DMA correlator uses (k_corr4d.h): reads by `buffer_load_dwordx4 ... lds` (16 bytes per
// lane), writes by 8-byte-per-lane stores.  Each kernel moves a KNOWN number of bytes from a
// buffer much larger than the caches; run it under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE ... / --pmc WRITE_SIZE ...
// and divide (MI355X_MICROARCH.md: FETCH_SIZE counts 128-byte requests at 64 for 16-byte-per-lane
// loads; other shapes "calibrate on a known byte count in your own access pattern").
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void dma16(v4i rsrc, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_dst)
                 : "memory");
}

// every workgroup copies `per_wg` bytes (a multiple of 4096): 16-byte-per-lane DMA in, 8-byte-per-lane stores out
__global__ __launch_bounds__(256) void k_copy_dma16_store8(const char* src, char* dst, unsigned per_wg)
{
    __shared__ __attribute__((aligned(16))) char buf[4096];
    const unsigned t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const char* s = src + (size_t)blockIdx.x * per_wg;
    char* d = dst + (size_t)blockIdx.x * per_wg;
    v4i rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)s);
    rs.y = __builtin_amdgcn_readfirstlane((int)(((size_t)s >> 32) & 0xffff));
    rs.z = (int)per_wg;
    rs.w = 0x00020000;
    for (unsigned off = 0; off < per_wg; off += 4096) {
        dma16(rs, off + wave * 1024u + lane * 16u, __builtin_amdgcn_readfirstlane((unsigned)(size_t)buf + wave * 1024u));
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        const v2u a = *(const v2u*)(buf + t * 8), b = *(const v2u*)(buf + 2048 + t * 8);
        *(v2u*)(d + off + t * 8) = a;
        *(v2u*)(d + off + 2048 + t * 8) = b;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// the calibrated shape of the guide: 16-byte loads and stores
__global__ __launch_bounds__(256) void k_copy_16_16(const float4* src, float4* dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        dst[i] = src[i];
}

// 8-byte loads and stores (the F = 2048 correlator's and k_corr4_main's shape)
__global__ __launch_bounds__(256) void k_copy_8_8(const float2* src, float2* dst, size_t n8)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256)
        dst[i] = src[i];
}

int main()
{
    const size_t bytes = (size_t)2 << 30; // 2 GiB each way, far beyond L2 / Infinity Cache
    char *s, *d;
    if (hipMalloc(&s, bytes) != hipSuccess || hipMalloc(&d, bytes) != hipSuccess)
        return 1;
    (void)hipMemset(s, 1, bytes);
    (void)hipMemset(d, 0, bytes);
    const unsigned nwg = 4096, per_wg = (unsigned)(bytes / nwg);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_copy_dma16_store8, dim3(nwg), dim3(256), 0, 0, s, d, per_wg);
        hipLaunchKernelGGL(k_copy_16_16, dim3(4096), dim3(256), 0, 0, (const float4*)s, (float4*)d, bytes / 16);
        hipLaunchKernelGGL(k_copy_8_8, dim3(4096), dim3(256), 0, 0, (const float2*)s, (float2*)d, bytes / 8);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("moved %zu bytes each way per launch: %s\n", bytes, hipGetErrorString(e));
    return e != hipSuccess;
}
