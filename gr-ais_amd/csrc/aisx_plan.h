// aisx_plan.h -- host-side set-up arithmetic shared by the C-ABI (aisx_lib.hip)
// and the CPU model in tests/emul: constructor maths of the reference blocks and
// launch geometry.  No device code.
#pragma once
#include <math.h>

#include <algorithm>
#include <vector>

#include "aisx_common.h"
#include "k_corr.h"
#include "k_corr4k.h"
#include "k_corr4d.h"
#include "k_corr4e.h"
#include "k_corr4f.h"
#include "k_corr2d.h"

namespace aisx {

struct CorrSetup {
    std::vector<cf> symbols; // d_symbols: reversed conjugate (lib/corr_est_cc_impl.cc:58-63)
    unsigned mark_delay;     // :65-66
    float thresh;            // :71-74
    int isps;                // :193
    int out_multiple;        // :77-85 nsamples of the reference's FFT filter
};

inline CorrSetup corr_setup(const cf* symbols, int nsym, float sps, unsigned mark_delay, float threshold)
{
    CorrSetup s;
    s.symbols.resize(nsym);
    for (int i = 0; i < nsym; i++)
        s.symbols[i] = cconj(symbols[nsym - 1 - i]);
    s.mark_delay = mark_delay >= (unsigned)nsym ? (unsigned)nsym - 1 : mark_delay;
    float corr = 0;
    for (int i = 0; i < nsym; i++)
        corr += cabs_f(cmul_exact(s.symbols[i], cconj(s.symbols[i])));
    s.thresh = threshold * corr * corr;
    s.isps = (int)(sps + 0.5f);
    int fftsize = (int)(2 * pow(2.0, ceil(log((double)nsym) / log(2.0))));
    s.out_multiple = fftsize - nsym + 1;
    return s;
}

// which FFT build serves a template of nsym samples: F = 2048 up to 512 samples (more
// valid outputs per flop than F = 4096 for short templates), F = 4096 up to 2048
inline int corr_pick_fft(int nsym) { return nsym <= CF_F / 4 ? CF_F : CF4_F; }
constexpr int CORR_MAX_TEMPLATE = CF4_F / 2;

inline std::vector<cf> corr_wtab(int F)
{
    std::vector<cf> w(F);
    for (int k = 0; k < F; k++) {
        double a = -2.0 * M_PI * (double)k / (double)F;
        w[k] = mk((float)cos(a), (float)sin(a));
    }
    return w;
}

// taps/F zero padded to F (fft_filter_ccc::set_taps scales the taps by 1/fftsize)
inline std::vector<cf> corr_padded_taps(const std::vector<cf>& stored, int F)
{
    std::vector<cf> pad(F, mk(0.f, 0.f));
    const float scale = 1.0f / (float)F;
    for (size_t i = 0; i < stored.size(); i++)
        pad[i] = mk(stored[i].re * scale, stored[i].im * scale);
    return pad;
}

// grid of the main kernel: (nseg, nchan) workgroups, each walking tiles_per_seg
// consecutive tiles of L outputs of one channel.  The segment count is chosen so
// that the grid is close to a whole number of full-chip rounds (256 CUs x 6
// resident workgroups of the F = 2048 build at 154 VGPRs / 18 KB LDS, 2 of the F = 4096 build): a ragged last round is pure loss
// for a kernel whose workgroups all take the same time.
inline void corr_grid(int nchan, int n, int L, int F, int* nseg, int* tiles_per_seg, int wg_per_cu = 0)
{
    const int ntiles = (n + L - 1) / L;
    // resident workgroups per CU: 6 (F = 2048), 3 (F = 4096, k_corr4k.h), 2 (F = 4096 with the
    // window prefetched by DMA, k_corr4d.h: two LDS images per workgroup)
    const long slots = 256L * (wg_per_cu > 0 ? wg_per_cu : (F == CF_F ? 6 : 3));
    int best = 1;
    double best_cost = 1e30;
    for (int ns = 1; ns <= 16; ns++) {
        const int tps = (ntiles + ns - 1) / ns;
        if (ns > 1 && tps < 6)
            break; // keep segments long: each one re-reads N samples of overlap
        const int real_ns = (ntiles + tps - 1) / tps;
        const long wgs = (long)nchan * real_ns;
        const long rounds = (wgs + slots - 1) / slots;
        // time ~ rounds * tiles per workgroup (+ a little for the extra overlap reads)
        const double cost = (double)rounds * tps * (1.0 + 0.01 * real_ns);
        if (cost < best_cost) {
            best_cost = cost;
            best = real_ns;
        }
    }
    const int tps = (ntiles + best - 1) / best;
    *nseg = (ntiles + tps - 1) / tps;
    *tiles_per_seg = tps;
}

struct MskSetup {
    float d_sps, gain_omega;
};
inline MskSetup msk_setup(float sps, float gain)
{
    MskSetup m;
    m.d_sps = (float)(sps / 2.0);                 // lib/msk_timing_recovery_cc_impl.cc:70
    m.gain_omega = (float)(gain * gain * 0.25);   // :83
    return m;
}

} // namespace aisx
