// chain_harness.cpp -- a compiled caller of the pipelined chain of the C ABI (include/aisx.h:
// aisx_chain_*), no Python in the process: what a C++ maintainer's replacement of the hier block
// python/ais_demod.py:21-56 does.  NCH channels of pseudo-random IQ with planted template bursts go
//   (a) through aisx_chain_step, STEPS steps issued back to back, the next step's input announced
//       one step ahead, two input buffers refilled in turn behind aisx_chain_wait_input, three
//       output sets in rotation behind aisx_chain_wait;
//   (b) through the four stage calls one after the other on the null stream, on handles of their own
//       (aisx_freqsync_process -> aisx_agc_process -> aisx_corr_process -> aisx_msk_process_stream),
// and every step's bits, symbol counts and tags must be the same, byte for byte.
//
//   g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I include tests/abi_cpp/chain_harness.cpp
//       -L gr-ais_amd/lib -laisx -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/gr-ais_amd/lib -o chain_harness
//
// TEST INFRASTRUCTURE (built and run by tests/test_abi_cpp.py under -m gpu).
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "aisx.h"

typedef aisx_cf32 cf;

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int rc__ = (call);                                                                       \
        if (rc__ < 0) {                                                                          \
            fprintf(stderr, "%s -> %d: %s (line %d)\n", #call, rc__, aisx_last_error(), __LINE__); \
            return 2;                                                                            \
        }                                                                                        \
    } while (0)
#define HIPCK(call)                                                                     \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s (line %d)\n", #call, hipGetErrorString(e__), __LINE__); \
            return 2;                                                                   \
        }                                                                               \
    } while (0)

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static double urand() // xorshift64*: deterministic, no library
{
    g_state ^= g_state >> 12;
    g_state ^= g_state << 25;
    g_state ^= g_state >> 27;
    return (double)((g_state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0;
}
static double nrand() { return sqrt(-2.0 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand()); }

// continuous-phase GMSK (BT 0.4, h = 1/2, Gaussian over four symbols) of a bit sequence at sps samples
// per symbol: what the chain is built for (its frequency estimator looks for the two spectral lines
// of the SQUARED signal, python/gmsk_sync.py:22-25).  Not GNU Radio's modulator to the last place --
// corr_est_cc takes its template as an argument (lib/corr_est_cc_impl.cc:48-63).
static std::vector<cf> gmsk(const std::vector<int>& bits, int sps, double phase0)
{
    const int nt = 4 * sps;
    std::vector<double> g(nt), q(nt + sps - 1, 0.0);
    double sum = 0;
    for (int i = 0; i < nt; i++) {
        const double t = (i - (nt - 1) / 2.0) / sps, a = 2.0 * 3.141592653589793 * 0.4 / sqrt(log(2.0));
        g[i] = exp(-0.5 * a * a * t * t);
        sum += g[i];
    }
    for (int i = 0; i < nt; i++)
        for (int k = 0; k < sps; k++)
            q[i + k] += g[i] / sum; // Gaussian (*) rectangle of one symbol: sums to sps
    const int n = (int)bits.size() * sps;
    std::vector<double> inc(n + (int)q.size(), 0.0);
    for (size_t b = 0; b < bits.size(); b++)
        for (size_t k = 0; k < q.size(); k++)
            inc[b * sps + k] += (bits[b] ? 1.0 : -1.0) * q[k] * (3.141592653589793 / 2.0 / sps);
    std::vector<cf> out(n);
    double ph = phase0;
    for (int i = 0; i < n; i++) {
        ph += inc[i];
        out[i] = cf{ (float)cos(ph), (float)sin(ph) };
    }
    return out;
}

struct Handles {
    aisx_freqsync* fs = nullptr;
    aisx_agc* agc = nullptr;
    aisx_corr* corr = nullptr;
    aisx_msk* msk = nullptr;
};

static int make_handles(Handles& h, const std::vector<cf>& tmpl, int nch, int T)
{
    const float sps = 4.f;
    CHECK(aisx_freqsync_create(&h.fs, sps * 9600.0, 9600.0, 1024, nch, T));           // python/ais_demod.py:34
    CHECK(aisx_agc_create(&h.agc, 512, 2.f, nch, T + 1024));                           // :35
    CHECK(aisx_corr_create(&h.corr, tmpl.data(), (int)tmpl.size(), sps, 1, 0.9f, nch, T + 1024, 1040)); // :39-42
    CHECK(aisx_msk_create(&h.msk, sps, 0.04f, 0.01f, 1, nch, T + 1024));              // :43-46
    return 0;
}

int main()
{
    const int NCH = 48, T = 16384, STEPS = 6, N = 896;
    int ndev = 0;
    if (aisx_device_count(&ndev) != AISX_OK || ndev < 1) {
        fprintf(stderr, "no device\n");
        return 3;
    }
    // the stock template: bytes [1, 1, 0, 0] * 7, most significant bit first (python/ais_demod.py:36-38)
    std::vector<int> pre;
    for (int r = 0; r < 7; r++)
        for (int byte : { 1, 1, 0, 0 })
            for (int b = 7; b >= 0; b--)
                pre.push_back((byte >> b) & 1);
    const std::vector<cf> tmpl = gmsk(pre, 4, 0.0);
    if ((int)tmpl.size() != N)
        return 2;
    std::vector<cf> x((size_t)STEPS * NCH * T);
    for (int s = 0; s < STEPS; s++)
        for (int c = 0; c < NCH; c++) {
            cf* row = &x[((size_t)s * NCH + c) * T];
            for (int i = 0; i < T; i++)
                row[i] = cf{ (float)(0.02 * nrand()), (float)(0.02 * nrand()) };
            for (int b = 0; b < 3; b++) { // three bursts per channel and step: the preamble, then 180 random bits
                std::vector<int> bits = pre;
                for (int k = 0; k < 180; k++)
                    bits.push_back(urand() < 0.5);
                const std::vector<cf> w = gmsk(bits, 4, 6.283185307179586 * urand());
                const int p0 = b * (T / 3) + (int)(urand() * (T / 3 - (int)w.size() - 8));
                const double cf0 = 2.0 * 3.141592653589793 * (urand() - 0.5) * 600.0 / 38400.0; // +-300 Hz
                for (size_t i = 0; i < w.size(); i++) {
                    const double cr = cos(cf0 * i), sr = sin(cf0 * i);
                    row[p0 + i].re += (float)(0.3 * (w[i].re * cr - w[i].im * sr));
                    row[p0 + i].im += (float)(0.3 * (w[i].re * sr + w[i].im * cr));
                }
            }
        }
    Handles A, B;
    if (make_handles(A, tmpl, NCH, T) || make_handles(B, tmpl, NCH, T))
        return 2;
    aisx_chain* chain = nullptr;
    CHECK(aisx_chain_create(&chain, A.fs, A.agc, A.corr, A.msk, NCH, T, 1024));
    const int cap = aisx_msk_out_capacity(A.msk);
    if (cap != aisx_msk_out_capacity(B.msk) || aisx_chain_depth() != AISX_CHAIN_DEPTH)
        return 2;
    hipStream_t st;
    HIPCK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    cf* d_in[2];
    uint8_t* d_bits[AISX_CHAIN_DEPTH];
    int* d_prod[AISX_CHAIN_DEPTH];
    for (int k = 0; k < 2; k++)
        HIPCK(hipMalloc((void**)&d_in[k], sizeof(cf) * (size_t)NCH * T));
    for (int k = 0; k < AISX_CHAIN_DEPTH; k++) {
        HIPCK(hipMalloc((void**)&d_bits[k], (size_t)NCH * cap));
        HIPCK(hipMalloc((void**)&d_prod[k], sizeof(int) * NCH));
    }
    std::vector<std::vector<uint8_t>> bitsA(STEPS, std::vector<uint8_t>((size_t)NCH * cap));
    std::vector<std::vector<int>> prodA(STEPS, std::vector<int>(NCH));
    std::vector<std::vector<aisx_tag>> tagsA(STEPS);
    const size_t step_items = (size_t)NCH * T;

    // ---- (a) the pipelined chain
    HIPCK(hipMemcpyAsync(d_in[0], &x[0], sizeof(cf) * step_items, hipMemcpyHostToDevice, st));
    for (int s = 0; s < STEPS; s++) {
        if (s + 1 < STEPS) { // block s + 1 arrives in the other buffer, whose last reader was step s - 1
            if (s >= 1)
                CHECK(aisx_chain_wait_input(chain, s - 1, st, 0));
            HIPCK(hipMemcpyAsync(d_in[(s + 1) & 1], &x[(size_t)(s + 1) * step_items], sizeof(cf) * step_items, hipMemcpyHostToDevice, st));
        }
        long long step = -1;
        const int o = s % AISX_CHAIN_DEPTH;
        CHECK(aisx_chain_step(chain, d_in[s & 1], T, T, s + 1 < STEPS ? d_in[(s + 1) & 1] : nullptr, T, T, nullptr, d_bits[o], cap,
                              d_prod[o], st, &step));
        if (step != s)
            return 2;
        // results of step s - 2 (still in flight: s - 1, s): read behind aisx_chain_wait
        const int r = s - 2;
        if (r >= 0) {
            CHECK(aisx_chain_wait(chain, r, st, 0));
            HIPCK(hipMemcpyAsync(bitsA[r].data(), d_bits[r % AISX_CHAIN_DEPTH], bitsA[r].size(), hipMemcpyDeviceToHost, st));
            HIPCK(hipMemcpyAsync(prodA[r].data(), d_prod[r % AISX_CHAIN_DEPTH], sizeof(int) * NCH, hipMemcpyDeviceToHost, st));
        }
    }
    for (int r = STEPS - 2; r < STEPS; r++) {
        if (r < 0)
            continue;
        CHECK(aisx_chain_wait(chain, r, nullptr, 1)); // (the host blocks)
        HIPCK(hipMemcpy(bitsA[r].data(), d_bits[r % AISX_CHAIN_DEPTH], bitsA[r].size(), hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(prodA[r].data(), d_prod[r % AISX_CHAIN_DEPTH], sizeof(int) * NCH, hipMemcpyDeviceToHost));
    }
    CHECK(aisx_chain_synchronize(chain));
    HIPCK(hipStreamSynchronize(st));
    for (int back = 0; back < 3 && back < STEPS; back++) { // the handle keeps the tags of its last three calls
        std::vector<aisx_tag>& tg = tagsA[STEPS - 1 - back];
        tg.resize((size_t)NCH * 1040);
        int nt = 0;
        CHECK(aisx_corr_read_tags_back(A.corr, back, tg.data(), (int)tg.size(), &nt, nullptr));
        tg.resize(nt);
    }

    // ---- (b) the stages one after the other, null stream, handles B
    cf *d_x, *d_y1, *d_y2, *d_y3;
    uint8_t* d_b;
    int* d_p;
    const long ys = T + 1024;
    HIPCK(hipMalloc((void**)&d_x, sizeof(cf) * step_items));
    HIPCK(hipMalloc((void**)&d_y1, sizeof(cf) * (size_t)NCH * ys));
    HIPCK(hipMalloc((void**)&d_y2, sizeof(cf) * (size_t)NCH * ys));
    HIPCK(hipMalloc((void**)&d_y3, sizeof(cf) * (size_t)NCH * ys));
    HIPCK(hipMalloc((void**)&d_b, (size_t)NCH * cap));
    HIPCK(hipMalloc((void**)&d_p, sizeof(int) * NCH));
    long nbits = 0, ntags = 0;
    int bad = 0;
    for (int s = 0; s < STEPS; s++) {
        HIPCK(hipMemcpy(d_x, &x[(size_t)s * step_items], sizeof(cf) * step_items, hipMemcpyHostToDevice));
        int m = 0;
        CHECK(aisx_freqsync_process(B.fs, d_x, T, T, d_y1, ys, nullptr, 0, &m, nullptr));
        if (m != T)
            return 2;
        CHECK(aisx_agc_process(B.agc, d_y1, ys, d_y2, ys, m, nullptr));
        CHECK(aisx_corr_process(B.corr, d_y2, ys, d_y3, ys, nullptr, 0, m, nullptr));
        const aisx_tag* dt;
        const int* dc;
        int tcap;
        CHECK(aisx_corr_tags_device(B.corr, &dt, &dc, &tcap));
        CHECK(aisx_msk_process_stream(B.msk, d_y3, ys, m, dt, dc, tcap, nullptr, nullptr, nullptr, d_b, cap, d_p, nullptr));
        std::vector<uint8_t> bb((size_t)NCH * cap);
        std::vector<int> pp(NCH);
        HIPCK(hipMemcpy(bb.data(), d_b, bb.size(), hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(pp.data(), d_p, sizeof(int) * NCH, hipMemcpyDeviceToHost));
        int st_msk = 0;
        CHECK(aisx_msk_last_status(B.msk, &st_msk, nullptr));
        if (st_msk != 0)
            bad++;
        for (int c = 0; c < NCH; c++) {
            if (pp[c] != prodA[s][c] || pp[c] < (T - 1024) / 4 - 64 || memcmp(&bb[(size_t)c * cap], &bitsA[s][(size_t)c * cap], pp[c]) != 0) {
                if (bad < 5)
                    fprintf(stderr, "step %d channel %d: %d symbols (chain: %d), bits differ or count off\n", s, c, pp[c], prodA[s][c]);
                bad++;
            }
            nbits += pp[c];
        }
        if (s >= STEPS - 3) {
            std::vector<aisx_tag> tb((size_t)NCH * 1040);
            int nt = 0;
            CHECK(aisx_corr_read_tags(B.corr, tb.data(), (int)tb.size(), &nt, nullptr));
            const std::vector<aisx_tag>& ta = tagsA[s];
            if ((size_t)nt != ta.size() || (nt > 0 && memcmp(tb.data(), ta.data(), sizeof(aisx_tag) * nt) != 0)) {
                fprintf(stderr, "step %d: tags differ (%d against %zu)\n", s, nt, ta.size());
                bad++;
            }
            ntags += nt;
        }
    }
    int st_a = 0;
    CHECK(aisx_msk_last_status(A.msk, &st_a, nullptr));
    printf("chain_harness: %d channels x %d steps x %d items: %ld bits and %ld tags compared, %d mismatches, status %d\n", NCH, STEPS, T,
           nbits, ntags, bad, st_a);
    CHECK(aisx_chain_destroy(chain));
    for (Handles* h : { &A, &B }) {
        aisx_freqsync_destroy(h->fs);
        aisx_agc_destroy(h->agc);
        aisx_corr_destroy(h->corr);
        aisx_msk_destroy(h->msk);
    }
    if (bad == 0 && st_a == 0 && ntags > 100 && nbits > (long)NCH * STEPS * (T / 4 - 400)) {
        printf("PASS\n");
        return 0;
    }
    printf("FAIL\n");
    return 1;
}
