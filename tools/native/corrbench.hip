// corrbench.hip -- torch-free A/B harness for the corr_est_cc path of libaisx.so builds:
//   corrbench <libaisx.so> [--ref <other libaisx.so>] [--nchan C] [--n T] [--N len] [--iters K] [--sps S]
// Fills [C][T] complex samples on the device (noise + template bursts), runs K timed
// aisx_corr_process calls through the C ABI of the library given, prints the main kernel's
// average duration (the library's own hipEvents) and the wall time per call; with --ref the
// same input goes through a second build and the outputs / tags are compared.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/aisx.h"

#define CK(e)                                                                        \
    do {                                                                             \
        hipError_t e__ = (e);                                                        \
        if (e__ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s (line %d)\n", #e, hipGetErrorString(e__), __LINE__); \
            exit(2);                                                                 \
        }                                                                            \
    } while (0)

struct Lib {
    void* h = nullptr;
    decltype(&aisx_corr_create) create;
    decltype(&aisx_corr_destroy) destroy;
    decltype(&aisx_corr_process) process;
    decltype(&aisx_corr_set_profiling) set_prof;
    decltype(&aisx_corr_kernel_ms_history) hist;
    decltype(&aisx_corr_read_tags) read_tags;
    decltype(&aisx_last_error) last_error;
    void open(const char* path)
    {
        h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            fprintf(stderr, "dlopen %s: %s\n", path, dlerror());
            exit(2);
        }
#define SYM(f, n) f = (decltype(f))dlsym(h, n); if (!f) { fprintf(stderr, "missing %s\n", n); exit(2); }
        SYM(create, "aisx_corr_create");
        SYM(destroy, "aisx_corr_destroy");
        SYM(process, "aisx_corr_process");
        SYM(set_prof, "aisx_corr_set_profiling");
        SYM(hist, "aisx_corr_kernel_ms_history");
        SYM(read_tags, "aisx_corr_read_tags");
        SYM(last_error, "aisx_last_error");
#undef SYM
    }
};

__device__ inline unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// noise (sum of four uniforms, sigma) + the template at amplitude 1 in pseudo-random slots
__global__ void k_fill(aisx_cf32* x, long stride, int T, const aisx_cf32* tmpl, int N, float sigma, int slot)
{
    const int c = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= T)
        return;
    const unsigned h0 = hash32((unsigned)c * 0x9e3779b9u + (unsigned)k * 2u + 1u), h1 = hash32(h0 + 0x68bc21ebu);
    auto uni4 = [](unsigned h) { return ((h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255) + (h >> 24)) * (1.0f / 255.0f) - 2.0f; };
    float re = sigma * 1.7320508f * uni4(h0), im = sigma * 1.7320508f * uni4(h1);
    const int s = k / slot, off = k - s * slot;
    const unsigned hs = hash32((unsigned)c * 7919u + (unsigned)s * 104729u + 12345u);
    const int start = (int)(hs % 97u);
    if ((hs >> 20) & 1u) {
        const int j = off - start;
        if (j >= 0 && j < N) {
            const float ph = (float)(hs & 1023u) * (6.2831853f / 1024.f);
            const float cs = cosf(ph), sn = sinf(ph);
            re += tmpl[j].re * cs - tmpl[j].im * sn;
            im += tmpl[j].re * sn + tmpl[j].im * cs;
        }
    }
    x[(long)c * stride + k].re = re;
    x[(long)c * stride + k].im = im;
}

int main(int argc, char** argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: corrbench lib.so [--ref ref.so] [--nchan C] [--n T] [--N len] [--iters K] [--sps S] [--calls-n n2]\n");
        return 2;
    }
    const char* libp = argv[1];
    const char* refp = nullptr;
    int nchan = 4096, T = 65536, N = 896, iters = 20;
    float sps = 4.f;
    for (int i = 2; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--ref")) refp = argv[i + 1];
        else if (!strcmp(argv[i], "--nchan")) nchan = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--n")) T = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--N")) N = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--iters")) iters = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--sps")) sps = (float)atof(argv[i + 1]);
    }
    // template: unit modulus, phase random walk of +-pi/2 per symbol (MSK-like)
    std::vector<aisx_cf32> tm(N);
    {
        double ph = 0;
        unsigned r = 12345;
        int isps = (int)(sps + 0.5f);
        double step = 0;
        for (int j = 0; j < N; j++) {
            if (j % isps == 0) {
                r = r * 1664525u + 1013904223u;
                step = ((r >> 16) & 1) ? M_PI / 2 / isps : -M_PI / 2 / isps;
            }
            ph += step;
            tm[j].re = (float)cos(ph);
            tm[j].im = (float)sin(ph);
        }
    }
    aisx_cf32 *d_x, *d_t, *d_o[2];
    const size_t bytes = sizeof(aisx_cf32) * (size_t)nchan * T;
    CK(hipMalloc(&d_x, bytes));
    CK(hipMalloc(&d_t, sizeof(aisx_cf32) * N));
    CK(hipMemcpy(d_t, tm.data(), sizeof(aisx_cf32) * N, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_o[0], bytes));
    if (refp)
        CK(hipMalloc(&d_o[1], bytes));
    hipLaunchKernelGGL(k_fill, dim3((T + 255) / 256, nchan), dim3(256), 0, 0, d_x, (long)T, T, d_t, N, 0.25f, 2 * N + 301);
    CK(hipDeviceSynchronize());

    const int tag_cap = 4 * 512;
    std::vector<aisx_tag> tags[2];
    int ntags[2] = { 0, 0 };
    const char* paths[2] = { libp, refp };
    for (int v = 0; v < (refp ? 2 : 1); v++) {
        // "--ref same": the reference run is the same library with AISX_CORR_DMA=0 (the k_corr4k.h build)
        if (refp && !strcmp(refp, "same")) {
            paths[1] = libp;
            if (v == 1)
                setenv("AISX_CORR_DMA", "0", 1);
        }
        Lib L;
        L.open(paths[v]);
        aisx_corr* h = nullptr;
        int rc = L.create(&h, tm.data(), N, sps, 1, 0.9f, nchan, T, tag_cap);
        if (rc != AISX_OK) {
            fprintf(stderr, "create: %d %s\n", rc, L.last_error());
            return 2;
        }
        L.set_prof(h, 1);
        // two calls so that the second starts from a carried history
        for (int k = 0; k < 2; k++) {
            rc = L.process(h, d_x, T, d_o[v], T, nullptr, 0, T, nullptr);
            if (rc != AISX_OK) {
                fprintf(stderr, "process: %d %s\n", rc, L.last_error());
                return 2;
            }
        }
        CK(hipDeviceSynchronize());
        tags[v].resize((size_t)nchan * 64);
        rc = L.read_tags(h, tags[v].data(), (int)tags[v].size(), &ntags[v], nullptr);
        L.set_prof(h, 1);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int k = 0; k < iters; k++)
            L.process(h, d_x, T, d_o[v], T, nullptr, 0, T, nullptr);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float wall;
        CK(hipEventElapsedTime(&wall, e0, e1));
        std::vector<float> ms(64);
        int nms = 0;
        L.hist(h, ms.data(), 64, &nms);
        double s = 0;
        float mn = 1e9f;
        for (int k = 0; k < nms; k++) {
            s += ms[k];
            mn = std::min(mn, ms[k]);
        }
        const double avg = nms ? s / nms : 0;
        if (getenv("CORRBENCH_SERIES")) {
            printf("series:");
            for (int k = 0; k < nms; k++)
                printf(" %.3f", ms[k]);
            printf("\n");
        }
        printf("%s: main kernel avg %.4f ms min %.4f (%d launches)  call wall %.4f ms  frac_of_8TB/s %.3f  tags(read rc %d) %d\n",
               paths[v], avg, mn, nms, wall / iters, avg > 0 ? 16.0 * nchan * T / (avg * 1e-3) / 8e12 : 0, rc, ntags[v]);
        if (auto prof = (int (*)(unsigned long long*, int))dlsym(L.h, "aisx_debug_ce_prof")) {
            unsigned long long v[16];
            prof(v, 1);
            const char* nm[8] = { "wait window", "barrier top", "load/store/next", "pass 1", "barrier 1", "passes 2-4,H,4-2", "barrier 2", "last pass+thresh" };
            double tot = 0;
            for (int i = 0; i < 8; i++)
                tot += (double)v[i];
            printf("section timers (wave-ticks of 10 ns, %llu wave-tiles):", v[8]);
            for (int i = 0; i < 8; i++)
                printf("  %s %.1f%% (%.0f ns/tile)", nm[i], 100.0 * v[i] / tot, 10.0 * v[i] / (double)v[8]);
            printf("\n");
        }
        L.destroy(h);
    }
    if (refp) {
        // pass-through bit for bit, tags: same count, offsets equal, values within 1e-5 relative
        std::vector<aisx_cf32> a((size_t)T), b((size_t)T);
        long bad = 0;
        for (int c = 0; c < nchan; c += std::max(1, nchan / 64)) {
            CK(hipMemcpy(a.data(), d_o[0] + (size_t)c * T, sizeof(aisx_cf32) * T, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), d_o[1] + (size_t)c * T, sizeof(aisx_cf32) * T, hipMemcpyDeviceToHost));
            bad += memcmp(a.data(), b.data(), sizeof(aisx_cf32) * T) != 0;
        }
        long tbad = 0;
        double worst = 0, worst_t = 0;
        if (ntags[0] != ntags[1])
            tbad = -1;
        else
            for (int k = 0; k < ntags[0]; k++) {
                const aisx_tag &p = tags[0][k], &q = tags[1][k];
                if (p.offset != q.offset || p.key != q.key || p.chan != q.chan) {
                    tbad++;
                    continue;
                }
                const double d = fabs(p.value - q.value);
                if (p.key == AISX_KEY_TIME_EST || p.key == AISX_KEY_PHASE_EST)
                    worst_t = std::max(worst_t, d);
                else
                    worst = std::max(worst, d / std::max(1e-30, fabs(q.value)));
            }
        printf("compare: pass-through rows differing %ld (of 64 sampled), tags %d vs %d, mismatched %ld, worst mag rel %.3g, worst time/phase abs %.3g\n",
               bad, ntags[0], ntags[1], tbad, worst, worst_t);
    }
    return 0;
}
