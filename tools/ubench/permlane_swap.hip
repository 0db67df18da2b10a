#include <hip/hip_runtime.h>
__global__ void k(float* o, const float* in)
{
    float x = in[threadIdx.x];
    unsigned u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    o[threadIdx.x] = __uint_as_float(r[0]);
    o[64 + threadIdx.x] = __uint_as_float(r[1]);
    auto r2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    o[128 + threadIdx.x] = __uint_as_float(r2[0]);
    o[192 + threadIdx.x] = __uint_as_float(r2[1]);
}
int main()
{
    float *d, *di, h[256], hi[64];
    for (int i = 0; i < 64; i++) hi[i] = i;
    hipMalloc(&d, 1024); hipMalloc(&di, 256);
    hipMemcpy(di, hi, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, di);
    hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 256; i++) printf("%g%c", h[i], (i % 16 == 15) ? '\n' : ' ');
    return 0;
}
