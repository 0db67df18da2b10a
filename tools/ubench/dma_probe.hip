// dma_probe.hip -- what `buffer_load_dwordx4 ... offen lds` (LDS-DMA) does on gfx950, as the
// correlator's window prefetch uses it: lane-linear destination (M0 base + 16 B x lane),
// out-of-range lanes of the buffer resource write zeros, 8-byte-aligned sources.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(v4i rsrc, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_dst)
                 : "memory");
}

__global__ void k_probe(const float* src, unsigned nbytes, unsigned start_byte, float* out, unsigned lds_shift)
{
    extern __shared__ __attribute__((aligned(16))) char smem0[];
    char* smem = smem0 + lds_shift; // (destinations beyond 64 KiB: M0 must carry more than 16 address bits)
    const int t = threadIdx.x;
    // poison
    for (int i = t; i < 4096; i += blockDim.x)
        ((float*)smem)[i] = -7.f;
    __syncthreads();
    // buffer resource: base, stride 0, num_records = nbytes, raw 32-bit data format
    v4i rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)src);
    rs.y = __builtin_amdgcn_readfirstlane((int)(((size_t)src >> 32) & 0xffff));
    rs.z = __builtin_amdgcn_readfirstlane((int)nbytes);
    rs.w = 0x00020000;
    const unsigned wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned lane = t & 63;
    // wave w loads pieces w and w + 4 (1 KiB each) to LDS at piece * 1088 (a padded pitch)
    for (unsigned p = wave; p < 8; p += 4) {
        const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem + p * 1088u);
        dma16(rs, start_byte + p * 1024u + lane * 16u, lds_dst);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = t; i < 8 * 272; i += blockDim.x)
        out[i] = ((float*)smem)[i];
}

int main()
{
    const int N = 4096;
    std::vector<float> h(N);
    for (int i = 0; i < N; i++)
        h[i] = (float)i;
    float *d, *o;
    (void)hipMalloc(&d, N * 4);
    (void)hipMalloc(&o, 8 * 272 * 4);
    (void)hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    std::vector<float> r(8 * 272);
    const unsigned starts[3] = { 0, 8, 4 };
    for (int s = 0; s < 3; s++) {
        // the resource ends 100 floats before the last piece ends
        const unsigned nbytes = (unsigned)((8 * 256 - 100) * 4) + starts[s];
        const unsigned shift = (s == 2) ? 100u * 1024u : 0u;
        (void)hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(256), 16384 + shift, 0, d, nbytes, starts[s], o, shift);
        hipError_t e = hipDeviceSynchronize();
        (void)hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0, zeros = 0, poison = 0;
        for (int p = 0; p < 8; p++)
            for (int k = 0; k < 272; k++) {
                const float v = r[p * 272 + k];
                if (k >= 256) { poison += (v == -7.f); continue; }
                const int srcf = starts[s] / 4 + p * 256 + k;
                const bool inr = (unsigned)(srcf * 4 + 4) <= nbytes;
                if (inr) bad += (v != (float)srcf);
                else { zeros += (v == 0.f); bad += (v != 0.f); }
            }
        printf("lds shift %u: ", shift);
        printf("start %u B: err=%d  bad=%d  oob-zeros=%d (expect %d)  pad-untouched=%d (expect 128)\n", starts[s], (int)e, bad, zeros,
               100 - 0, poison);
    }
    return 0;
}
