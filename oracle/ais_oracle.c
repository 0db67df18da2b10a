/*
 * ais_oracle.c -- CPU restatement of the gr-ais demod hot path (see ais_oracle.h).
 * TEST INFRASTRUCTURE ONLY.  "parity unpinned" (no reference tests exist).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no fused multiply-add, so
 * that the float sequences below are the ones the HIP kernels reproduce).
 *
 * Reference citations are paths under /root/reference.  [GR] marks GNU Radio
 * 3.8 / VOLK code that is NOT in /root/reference (third-party dependency,
 * CMakeLists.txt:71 "Gnuradio 3.8", no lockfile) and is restated from its
 * published source.
 */
#include "ais_oracle.h"
#include "orc_tables.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ */
/* complex helpers: the formulas libstdc++ std::complex<float> uses    */
/* (a*c - b*d, a*d + b*c), each op rounded to float.                   */
/* ------------------------------------------------------------------ */
static inline orc_cf cmul(orc_cf a, orc_cf b)
{
    orc_cf r;
    r.re = a.re * b.re - a.im * b.im;
    r.im = a.re * b.im + a.im * b.re;
    return r;
}
static inline orc_cf cconj(orc_cf a)
{
    orc_cf r = { a.re, -a.im };
    return r;
}
static inline orc_cf cadd(orc_cf a, orc_cf b)
{
    orc_cf r = { a.re + b.re, a.im + b.im };
    return r;
}
static inline orc_cf csub(orc_cf a, orc_cf b)
{
    orc_cf r = { a.re - b.re, a.im - b.im };
    return r;
}
/* std::abs(std::complex<float>) = hypotf.  glibc evaluates hypotf as
 * (float)sqrt((double)x*x + (double)y*y) for finite arguments; written out so
 * that the HIP kernels can run the identical double sequence. */
static inline float cabs_f(orc_cf a)
{
    return (float)sqrt((double)a.re * (double)a.re + (double)a.im * (double)a.im);
}

/* ------------------------------------------------------------------ */
/* Work buffers.  A GNU Radio flowgraph allocates its block buffers    */
/* once, before it runs; the step functions below want theirs per call */
/* (and the timing harness builds a fresh chain per channel).  Freed   */
/* work buffers are therefore kept, per thread, for the next request   */
/* of that size: no malloc / mmap / page faults inside a timed loop    */
/* after the first channel.  Arithmetic is not affected.               */
/* ------------------------------------------------------------------ */
#define ORC_POOL_SLOTS 32
typedef struct { size_t cap, pad; } orc_pool_hdr; /* 16 bytes: user memory stays 16-byte aligned */
static __thread struct { orc_pool_hdr *blk[ORC_POOL_SLOTS]; } orc_pool;

static void *orc_big_alloc(size_t bytes)
{
    int best = -1;
    for (int i = 0; i < ORC_POOL_SLOTS; i++) {
        orc_pool_hdr *b = orc_pool.blk[i];
        if (b && b->cap >= bytes && (best < 0 || b->cap < orc_pool.blk[best]->cap))
            best = i;
    }
    if (best >= 0) {
        orc_pool_hdr *b = orc_pool.blk[best];
        orc_pool.blk[best] = NULL;
        return (void *)(b + 1);
    }
    orc_pool_hdr *b = (orc_pool_hdr *)malloc(sizeof(orc_pool_hdr) + (bytes ? bytes : 1));
    if (!b)
        return NULL;
    b->cap = bytes;
    b->pad = 0;
    return (void *)(b + 1);
}
static void orc_big_free(void *p)
{
    if (!p)
        return;
    orc_pool_hdr *b = (orc_pool_hdr *)p - 1;
    int smallest = -1;
    for (int i = 0; i < ORC_POOL_SLOTS; i++) {
        if (!orc_pool.blk[i]) {
            orc_pool.blk[i] = b;
            return;
        }
        if (smallest < 0 || orc_pool.blk[i]->cap < orc_pool.blk[smallest]->cap)
            smallest = i;
    }
    if (orc_pool.blk[smallest]->cap < b->cap) { /* full: keep the larger blocks */
        free(orc_pool.blk[smallest]);
        orc_pool.blk[smallest] = b;
    } else {
        free(b);
    }
}
/* gives this thread's kept buffers back to the C library */
void orc_pool_release(void)
{
    for (int i = 0; i < ORC_POOL_SLOTS; i++) {
        free(orc_pool.blk[i]);
        orc_pool.blk[i] = NULL;
    }
}

/* ------------------------------------------------------------------ */
/* [GR] gnuradio-runtime/lib/math/fast_atan2f.cc                       */
/* ------------------------------------------------------------------ */
float orc_fast_atan2f(float y, float x)
{
    const float TAN_MAP_RES = 0.003921569f; /* 1/255 */
    const int TAN_MAP_SIZE = 255;
    float x_abs, y_abs, z, alpha, angle, base_angle;
    int index;

    y_abs = fabsf(y);
    x_abs = fabsf(x);
    if (!((y_abs > 0.0f) || (x_abs > 0.0f)))
        return 0.0f;
    if (y_abs < x_abs)
        z = y_abs / x_abs;
    else
        z = x_abs / y_abs;
    if (z < TAN_MAP_RES) {
        base_angle = z;
    } else {
        alpha = z * (float)TAN_MAP_SIZE;
        index = ((int)alpha) & 0xff;
        alpha -= (float)index;
        base_angle = orc_atan_table[index];
        base_angle += (orc_atan_table[index + 1] - orc_atan_table[index]) * alpha;
    }
    if (x_abs > y_abs) {
        if (x >= 0.0f) {
            if (y >= 0.0f)
                angle = base_angle;
            else
                angle = -base_angle;
        } else {
            angle = 3.14159265358979323846f;
            if (y >= 0.0f)
                angle -= base_angle;
            else
                angle = base_angle - angle;
        }
    } else {
        if (y >= 0.0f) {
            angle = 1.57079632679489661923f;
            if (x >= 0.0f)
                angle -= base_angle;
            else
                angle += base_angle;
        } else {
            angle = -1.57079632679489661923f;
            if (x >= 0.0f)
                angle += base_angle;
            else
                angle -= base_angle;
        }
    }
    return angle;
}

/* ------------------------------------------------------------------ */
/* [GR] gr-filter mmse_fir_interpolator_cc::interpolate: row           */
/* rint(mu*128) of the 129x8 table, fir_filter_ccf (taps stored        */
/* reversed) evaluated with the VOLK generic dot product               */
/* (volk_32fc_32f_dot_prod_32fc_generic: sequential sum from index 0). */
/* y = sum_k taps[row][7-k] * in[k].                                   */
/* ------------------------------------------------------------------ */
orc_cf orc_mmse_interpolate(const orc_cf *in, float mu, int *err)
{
    int imu = (int)rint(mu * ORC_MMSE_NSTEPS);
    orc_cf acc = { 0.0f, 0.0f };
    if (imu < 0 || imu > ORC_MMSE_NSTEPS) { /* upstream throws std::runtime_error */
        if (err)
            *err = 1;
        return acc;
    }
    for (int k = 0; k < ORC_MMSE_NTAPS; k++) {
        float t = orc_mmse_taps[imu][ORC_MMSE_NTAPS - 1 - k];
        acc.re += in[k].re * t;
        acc.im += in[k].im * t;
    }
    return acc;
}

/* [GR] gnuradio/math.h gr::branchless_clip */
float orc_branchless_clip(float x, float clip)
{
    float x1 = fabsf(x + clip);
    float x2 = fabsf(x - clip);
    x1 -= x2;
    return (float)(0.5 * x1);
}

/* ------------------------------------------------------------------ */
/* [GR] gr::fxpt (gnuradio-runtime include/gnuradio/fxpt.h, 3.7 / 3.8): */
/* the fixed-point sin/cos GNU Radio's NCOs and frequency_modulator_fc   */
/* use.  32-bit angle (2^31 = pi), the top 10 bits pick a {slope, offset}*/
/* pair of s_sine_table, the line is evaluated at ux >> 1 in float.      */
/* The table is regenerated by tools/gen_tables.py from the recipe of    */
/* upstream's gen_sine_table.py (orc_tables.h); its first entry equals   */
/* upstream's first line digit for digit.                                */
/* ------------------------------------------------------------------ */
int32_t orc_fxpt_float_to_fixed(float x)
{
    const float PI = 3.14159265358979323846f, TAU = 2.0f * 3.14159265358979323846f, TWO_TO_THE_31 = 2147483648.0f;
    /* Fold x into -PI to PI. */
    int d = (int)floor(x / TAU + 0.5);
    x -= d * TAU;
    /* And convert to an integer. */
    return (int32_t)((float)x * TWO_TO_THE_31 / PI);
}

void orc_fxpt_float_to_fixed_n(const float *x, int32_t *out, long n)
{
    for (long i = 0; i < n; i++) out[i] = orc_fxpt_float_to_fixed(x[i]);
}

void orc_fxpt_sincos(int32_t x, float *s, float *c)
{
    uint32_t ux = (uint32_t)x;
    int sin_index = ux >> (32 - 10);
    *s = orc_sine_table[sin_index][0] * (ux >> 1) + orc_sine_table[sin_index][1];
    ux = (uint32_t)x + 0x40000000u;
    int cos_index = ux >> (32 - 10);
    *c = orc_sine_table[cos_index][0] * (ux >> 1) + orc_sine_table[cos_index][1];
}

/* frequency_modulator_fc's sin/cos of d_phase ([GR] frequency_modulator_fc_impl.cc:
 * angle = gr::fxpt::float_to_fixed(d_phase); gr::fxpt::sincos(angle, &oq, &oi)) */
void orc_nco_sincos(float phase, float *s, float *c)
{
    orc_fxpt_sincos(orc_fxpt_float_to_fixed(phase), s, c);
}

/* ------------------------------------------------------------------ */
/* Single-precision FFT standing in for FFTW (gr::fft::fft_complex).   */
/* Iterative radix-2 DIT, twiddles rounded from double.                */
/* ------------------------------------------------------------------ */
#define ORC_MAX_FFT_CACHE 8
static struct { int n; orc_cf *tw; int *rev; } g_fft_cache[ORC_MAX_FFT_CACHE];

/* the cpu_baseline leg of bench.py runs channels on several threads: entries are
 * created under a lock and published by writing .n last; nothing is evicted while
 * fewer than ORC_MAX_FFT_CACHE sizes are in use (the chain uses three) */
static pthread_mutex_t g_fft_lock = PTHREAD_MUTEX_INITIALIZER;
static int fft_cache_fill(int n);
static int fft_cache_get(int n)
{
    for (int i = 0; i < ORC_MAX_FFT_CACHE; i++)
        if (__atomic_load_n(&g_fft_cache[i].n, __ATOMIC_ACQUIRE) == n)
            return i;
    pthread_mutex_lock(&g_fft_lock);
    int i = fft_cache_fill(n);
    pthread_mutex_unlock(&g_fft_lock);
    return i;
}

static int fft_cache_fill(int n)
{
    int i;
    for (i = 0; i < ORC_MAX_FFT_CACHE; i++)
        if (g_fft_cache[i].n == n)
            return i;
    for (i = 0; i < ORC_MAX_FFT_CACHE; i++)
        if (g_fft_cache[i].n == 0)
            break;
    if (i == ORC_MAX_FFT_CACHE) {
        i = 0;
        g_fft_cache[0].n = 0;
        free(g_fft_cache[0].tw);
        free(g_fft_cache[0].rev);
    }
    g_fft_cache[i].tw = (orc_cf *)malloc(sizeof(orc_cf) * (n / 2 + 1));
    g_fft_cache[i].rev = (int *)malloc(sizeof(int) * n);
    for (int k = 0; k < n / 2; k++) {
        double a = -2.0 * M_PI * (double)k / (double)n;
        g_fft_cache[i].tw[k].re = (float)cos(a);
        g_fft_cache[i].tw[k].im = (float)sin(a);
    }
    int bits = 0;
    while ((1 << bits) < n)
        bits++;
    for (int k = 0; k < n; k++) {
        int r = 0;
        for (int b = 0; b < bits; b++)
            if (k & (1 << b))
                r |= 1 << (bits - 1 - b);
        g_fft_cache[i].rev[k] = r;
    }
    __atomic_store_n(&g_fft_cache[i].n, n, __ATOMIC_RELEASE);
    return i;
}

void orc_fft(orc_cf *buf, int n, int inverse)
{
    int ci = fft_cache_get(n);
    const orc_cf *tw = g_fft_cache[ci].tw;
    const int *rev = g_fft_cache[ci].rev;
    for (int k = 0; k < n; k++) {
        int r = rev[k];
        if (r > k) {
            orc_cf t = buf[k];
            buf[k] = buf[r];
            buf[r] = t;
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1, step = n / len;
        for (int i = 0; i < n; i += len) {
            for (int j = 0; j < half; j++) {
                orc_cf w = tw[j * step];
                if (inverse)
                    w.im = -w.im;
                orc_cf u = buf[i + j];
                orc_cf v = cmul(buf[i + j + half], w);
                buf[i + j] = cadd(u, v);
                buf[i + j + half] = csub(u, v);
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* [GR] gr-filter/lib/fft_filter.cc  kernel::fft_filter_ccc            */
/* (overlap-add; used at lib/corr_est_cc_impl.cc:77,84,144,188)        */
/* ------------------------------------------------------------------ */
typedef struct {
    int ntaps, fftsize, nsamples;
    orc_cf *xformed_taps, *tail, *fwd, *inv;
} fft_filter;

static void fft_filter_free(fft_filter *f)
{
    free(f->xformed_taps);
    free(f->tail);
    free(f->fwd);
    free(f->inv);
    memset(f, 0, sizeof(*f));
}

/* compute_sizes + set_taps; returns nsamples.  The tail is zeroed, as upstream. */
static int fft_filter_set_taps(fft_filter *f, const orc_cf *taps, int ntaps)
{
    fft_filter_free(f);
    f->ntaps = ntaps;
    f->fftsize = (int)(2 * pow(2.0, ceil(log((double)ntaps) / log(2.0))));
    f->nsamples = f->fftsize - ntaps + 1;
    f->xformed_taps = (orc_cf *)calloc(f->fftsize, sizeof(orc_cf));
    f->tail = (orc_cf *)calloc(ntaps > 1 ? ntaps - 1 : 1, sizeof(orc_cf));
    f->fwd = (orc_cf *)calloc(f->fftsize, sizeof(orc_cf));
    f->inv = (orc_cf *)calloc(f->fftsize, sizeof(orc_cf));
    float scale = 1.0f / f->fftsize;
    for (int i = 0; i < ntaps; i++) {
        f->xformed_taps[i].re = taps[i].re * scale;
        f->xformed_taps[i].im = taps[i].im * scale;
    }
    orc_fft(f->xformed_taps, f->fftsize, 0);
    return f->nsamples;
}

/* filter(): upstream requires nitems to be a multiple of nsamples (that is why
 * corr_est_cc sets the output multiple).  EXTENSION for the batched harness:
 * a final partial block of r < nsamples items is handled by the general
 * overlap-add rule (output r items, carry the remaining ntaps-1 partial sums),
 * which is the same linear convolution. */
static void fft_filter_filter(fft_filter *f, int nitems, const orc_cf *input, orc_cf *output)
{
    const int tailsize = f->ntaps - 1;
    for (int i = 0; i < nitems; i += f->nsamples) {
        int r = nitems - i < f->nsamples ? nitems - i : f->nsamples;
        memcpy(f->fwd, &input[i], sizeof(orc_cf) * r);
        memset(&f->fwd[r], 0, sizeof(orc_cf) * (f->fftsize - r));
        orc_fft(f->fwd, f->fftsize, 0);
        for (int j = 0; j < f->fftsize; j++) /* volk_32fc_x2_multiply_32fc */
            f->inv[j] = cmul(f->fwd[j], f->xformed_taps[j]);
        orc_fft(f->inv, f->fftsize, 1);
        for (int j = 0; j < tailsize; j++)
            f->inv[j] = cadd(f->inv[j], f->tail[j]);
        memcpy(&output[i], f->inv, sizeof(orc_cf) * r);
        memcpy(f->tail, &f->inv[r], sizeof(orc_cf) * tailsize);
    }
}

/* ------------------------------------------------------------------ */
/* corr_est_cc  (lib/corr_est_cc_impl.cc)                              */
/* ------------------------------------------------------------------ */
struct orc_corr {
    orc_cf *symbols; /* d_symbols */
    int nsym;
    float sps;            /* d_sps */
    unsigned mark_delay;  /* d_mark_delay */
    float thresh;         /* d_thresh */
    fft_filter filter;    /* d_filter */
    int history;          /* history() */
    int output_multiple;
    orc_cf *corr;    /* d_corr */
    float *corr_mag; /* d_corr_mag */
    int cap;
};

static void corr_reserve(orc_corr *h, int n)
{
    if (n <= h->cap)
        return;
    orc_big_free(h->corr);
    orc_big_free(h->corr_mag);
    h->cap = n;
    h->corr = (orc_cf *)orc_big_alloc(sizeof(orc_cf) * n);
    h->corr_mag = (float *)orc_big_alloc(sizeof(float) * n);
}

/* constructor, lib/corr_est_cc_impl.cc:48-117 */
orc_corr *orc_corr_create(const orc_cf *symbols, int nsym, float sps, unsigned mark_delay, float threshold)
{
    orc_corr *h = (orc_corr *)calloc(1, sizeof(orc_corr));
    h->sps = sps;
    h->nsym = nsym;
    h->symbols = (orc_cf *)malloc(sizeof(orc_cf) * nsym);
    /* :58-63 time-reversed conjugate */
    for (int i = 0; i < nsym; i++)
        h->symbols[i] = cconj(symbols[nsym - 1 - i]);
    /* :65-66 */
    h->mark_delay = mark_delay >= (unsigned)nsym ? (unsigned)nsym - 1 : mark_delay;
    /* :71-74  corr += abs(s*conj(s)) ; std::abs(complex<float>) = hypotf */
    float corr = 0;
    for (int i = 0; i < nsym; i++) {
        orc_cf p = cmul(h->symbols[i], cconj(h->symbols[i]));
        corr += cabs_f(p);
    }
    h->thresh = threshold * corr * corr;
    /* :77-85 */
    h->output_multiple = fft_filter_set_taps(&h->filter, h->symbols, nsym);
    /* :95 */
    h->history = nsym + 1;
    corr_reserve(h, 24 * 1024); /* :111-116 */
    return h;
}

void orc_corr_destroy(orc_corr *h)
{
    if (!h)
        return;
    fft_filter_free(&h->filter);
    free(h->symbols);
    orc_big_free(h->corr);
    orc_big_free(h->corr_mag);
    free(h);
}

int orc_corr_history(const orc_corr *h) { return h->history; }
int orc_corr_output_multiple(const orc_corr *h) { return h->output_multiple; }
int orc_corr_fftsize(const orc_corr *h) { return h->filter.fftsize; }
float orc_corr_threshold(const orc_corr *h) { return h->thresh; }
unsigned orc_corr_mark_delay(const orc_corr *h) { return h->mark_delay; }
void orc_corr_taps(const orc_corr *h, orc_cf *out) { memcpy(out, h->symbols, sizeof(orc_cf) * h->nsym); }

/* set_symbols, lib/corr_est_cc_impl.cc:132-162.  Quirk preserved: the taps are
 * stored WITHOUT conjugate/reverse and d_thresh is not recomputed. */
void orc_corr_set_symbols(orc_corr *h, const orc_cf *symbols, int nsym)
{
    free(h->symbols);
    h->nsym = nsym;
    h->symbols = (orc_cf *)malloc(sizeof(orc_cf) * nsym);
    memcpy(h->symbols, symbols, sizeof(orc_cf) * nsym);
    h->output_multiple = fft_filter_set_taps(&h->filter, h->symbols, nsym);
    h->history = nsym + 1;
    h->mark_delay = h->mark_delay >= (unsigned)nsym ? (unsigned)nsym - 1 : h->mark_delay;
}

static void add_tag(orc_tag *tags, int max_tags, int *ntags, int port, uint64_t off, int key, double val)
{
    if (*ntags < max_tags) {
        tags[*ntags].offset = off;
        tags[*ntags].key = key;
        tags[*ntags].value = val;
        tags[*ntags].port = port;
    }
    (*ntags)++;
}

/* work(), lib/corr_est_cc_impl.cc:164-279 */
int orc_corr_work(orc_corr *h, int noutput_items, const orc_cf *in, orc_cf *out, orc_cf *corr_out,
                  uint64_t nitems_written, orc_tag *tags, int max_tags, int *ntags)
{
    orc_cf *corr;
    *ntags = 0;
    corr_reserve(h, noutput_items);
    corr = corr_out ? corr_out : h->corr; /* :174-177 */
    unsigned hist_len = h->history - 1;   /* :180 */
    memcpy(out, &in[0], sizeof(orc_cf) * noutput_items); /* :184 */
    fft_filter_filter(&h->filter, noutput_items, &in[hist_len], corr); /* :188 */
    /* :191 volk_32fc_magnitude_squared_32f (generic: re*re + im*im) */
    for (int k = 0; k < noutput_items; k++)
        h->corr_mag[k] = corr[k].re * corr[k].re + corr[k].im * corr[k].im;

    int isps = (int)(h->sps + 0.5f); /* :193 */
    int i = 0;
    const float *mag = h->corr_mag;
    while (i < noutput_items) {
        if (mag[i] <= h->thresh) { /* :197-200 */
            i++;
            continue;
        }
        while ((i < (noutput_items - 1)) && (mag[i] < mag[i + 1])) /* :202-204 */
            i++;
        add_tag(tags, max_tags, ntags, 0, nitems_written + i, ORC_KEY_CORR_START, (double)mag[i]); /* :213 */
        double center = 0.0; /* :219-227 */
        if (i > 0 && i < (noutput_items - 1)) {
            double nom = 0, den = 0;
            for (int s = 0; s < 3; s++) {
                nom += (s + 1) * mag[i + s - 1];
                den += mag[i + s - 1];
            }
            center = nom / den - 2.0;
        }
        float phase = orc_fast_atan2f(corr[i].im, corr[i].re); /* :247 */
        int index = i + h->mark_delay;
        add_tag(tags, max_tags, ntags, 0, nitems_written + index, ORC_KEY_PHASE_EST, (double)phase);
        add_tag(tags, max_tags, ntags, 0, nitems_written + index, ORC_KEY_TIME_EST, center);
        add_tag(tags, max_tags, ntags, 0, nitems_written + index, ORC_KEY_CORR_EST, (double)mag[i]);
        if (corr_out) { /* :258-266 */
            add_tag(tags, max_tags, ntags, 1, nitems_written + i, ORC_KEY_PHASE_EST, (double)phase);
            add_tag(tags, max_tags, ntags, 1, nitems_written + i, ORC_KEY_TIME_EST, center);
            add_tag(tags, max_tags, ntags, 1, nitems_written + i, ORC_KEY_CORR_EST, (double)mag[i]);
        }
        i += isps; /* :270 */
    }
    return noutput_items;
}

/* ------------------------------------------------------------------ */
/* freqest  (lib/freqest_impl.cc)                                      */
/* ------------------------------------------------------------------ */
void orc_freqest_init(orc_freqest *f, float sample_rate, int data_rate, int fftlen)
{
    f->offset = (int)(fftlen * ((float)data_rate / (float)sample_rate)); /* :46 */
    f->binsize = (float)sample_rate / (float)fftlen;                      /* :47 */
    f->fftlen = fftlen;
}

/* work(), lib/freqest_impl.cc:57-88.  maxpos is NOT reset per vector (:68 vs :74). */
int orc_freqest_work(const orc_freqest *f, int noutput_items, const orc_cf *in, float *out)
{
    unsigned int fftlen = (unsigned)f->fftlen;
    float maxenergy = 0;
    unsigned int maxpos = 0;
    float currentenergy;
    for (int i = 0; i < noutput_items; i++) {
        maxenergy = 0;
        for (unsigned int j = 0; j < fftlen - f->offset; j++) {
            const orc_cf a = in[i * fftlen + j], b = in[i * fftlen + j + f->offset];
            currentenergy = cabs_f(a) + cabs_f(b); /* std::abs */
            if (currentenergy > maxenergy) {
                maxenergy = currentenergy;
                maxpos = j + f->offset / 2;
            }
        }
        out[i] = ((float)maxpos - fftlen / 2) * f->binsize / 2;
    }
    return noutput_items;
}

/* ------------------------------------------------------------------ */
/* square_and_fft_sync_cc  (python/gmsk_sync.py:14-37), pure wiring of */
/* [GR] multiply_cc, stream_to_vector, fft_vcc(forward, rectangular    */
/* window, shift), ais.freqest, repeat, frequency_modulator_fc,        */
/* multiply_cc.                                                        */
/* ------------------------------------------------------------------ */
struct orc_freqsync {
    orc_freqest fe;
    int fftlen;
    float sensitivity; /* frequency_modulator_fc(-1.0/(samplerate/(2*pi))) */
    float phase;       /* d_phase */
    orc_cf *pend;      /* stream_to_vector's partial vector */
    int npend;
    orc_cf *vecs;
    float *fhat;
    int cap_vec;
};

orc_freqsync *orc_freqsync_create(double samplerate, double bits_per_sec, int fftlen)
{
    orc_freqsync *h = (orc_freqsync *)calloc(1, sizeof(*h));
    /* gmsk_sync.py:25  ais.freqest(int(samplerate), int(bits_per_sec), fftlen) */
    orc_freqest_init(&h->fe, (float)(int)samplerate, (int)bits_per_sec, fftlen);
    h->fftlen = fftlen;
    h->sensitivity = (float)(-1.0 / (samplerate / (2 * M_PI))); /* gmsk_sync.py:27 */
    h->phase = 0;
    h->pend = (orc_cf *)calloc(fftlen, sizeof(orc_cf));
    return h;
}

void orc_freqsync_destroy(orc_freqsync *h)
{
    if (!h)
        return;
    free(h->pend);
    orc_big_free(h->vecs);
    orc_big_free(h->fhat);
    free(h);
}

int orc_freqsync_process(orc_freqsync *h, const orc_cf *in, int n, orc_cf *out, float *fhat_out)
{
    const int F = h->fftlen;
    int total = h->npend + n;
    int nvec = total / F;
    if (nvec > h->cap_vec) {
        orc_big_free(h->vecs);
        orc_big_free(h->fhat);
        h->cap_vec = nvec;
        h->vecs = (orc_cf *)orc_big_alloc(sizeof(orc_cf) * (size_t)nvec * F);
        h->fhat = (float *)orc_big_alloc(sizeof(float) * nvec);
    }
    /* assemble the stream (pending + new) that forms complete vectors */
    orc_cf *x = (orc_cf *)orc_big_alloc(sizeof(orc_cf) * (size_t)(nvec > 0 ? nvec * F : 1));
    int used = nvec * F;
    for (int k = 0; k < used; k++)
        x[k] = k < h->npend ? h->pend[k] : in[k - h->npend];
    /* square (multiply_cc of the stream with itself, gmsk_sync.py:22,30-31),
     * stream_to_vector, fft_vcc forward + fftshift (:23-24) */
    for (int v = 0; v < nvec; v++) {
        orc_cf *vec = &h->vecs[(size_t)v * F];
        for (int k = 0; k < F; k++)
            vec[k] = cmul(x[v * F + k], x[v * F + k]); /* window = 1.0: exact */
        orc_fft(vec, F, 0);
        for (int k = 0; k < F / 2; k++) { /* out[j] = X[(j + F/2) mod F] */
            orc_cf t = vec[k];
            vec[k] = vec[k + F / 2];
            vec[k + F / 2] = t;
        }
    }
    /* one freqest work() call over all the vectors (:25) */
    orc_freqest_work(&h->fe, nvec, h->vecs, h->fhat);
    /* repeat(fftlen) -> frequency_modulator_fc -> multiply_cc (:26-28,33) */
    for (int v = 0; v < nvec; v++) {
        if (fhat_out)
            fhat_out[v] = h->fhat[v];
        for (int k = 0; k < F; k++) {
            /* [GR] frequency_modulator_fc_impl::work */
            h->phase = h->phase + h->sensitivity * h->fhat[v];
            const float F_PI = (float)M_PI;
            h->phase = fmodf(h->phase + F_PI, 2.0f * F_PI) - F_PI;
            float s, c;
            orc_nco_sincos(h->phase, &s, &c);
            orc_cf nco = { c, s };
            out[v * F + k] = cmul(x[v * F + k], nco);
        }
    }
    /* keep the trailing partial vector */
    int rem = total - used;
    orc_cf *np = (orc_cf *)malloc(sizeof(orc_cf) * (size_t)(rem > 0 ? rem : 1));
    for (int k = 0; k < rem; k++) {
        int idx = used + k;
        np[k] = idx < h->npend ? h->pend[idx] : in[idx - h->npend];
    }
    memcpy(h->pend, np, sizeof(orc_cf) * rem);
    h->npend = rem;
    free(np);
    orc_big_free(x);
    return used;
}

/* ------------------------------------------------------------------ */
/* [GR] gr-analog feedforward_agc_cc_impl::work (python/ais_demod.py:35) */
/* ------------------------------------------------------------------ */
static inline float agc_envelope(orc_cf x)
{
    float r_abs = fabsf(x.re);
    float i_abs = fabsf(x.im);
    if (r_abs > i_abs)
        return (float)(r_abs + 0.4 * i_abs);
    else
        return (float)(i_abs + 0.4 * r_abs);
}

/* [GR] feedforward_agc_cc_impl::work as published in GNU Radio 3.7 / 3.8:
 *     // float max_env = 1e-12;   // avoid divide by zero
 *     float max_env = 1e-4;       // avoid divide by zero, indirectly set max gain
 * i.e. the live floor is 1e-4 (ORC_AGC_FLOOR); the older 1e-12 is kept reachable
 * through orc_feedforward_agc_floor for the tests.  The two differ only where a
 * whole window's envelope maximum lies in (0, 1e-4). */
void orc_feedforward_agc_floor(int nsamples, float reference, float floor_env, int noutput_items, const orc_cf *in,
                               orc_cf *out)
{
    for (int i = 0; i < noutput_items; i++) {
        float max_env = floor_env;
        for (int j = 0; j < nsamples; j++) {
            float e = agc_envelope(in[i + j]);
            max_env = max_env < e ? e : max_env; /* std::max(max_env, e) */
        }
        float gain = reference / max_env;
        out[i].re = gain * in[i].re;
        out[i].im = gain * in[i].im;
    }
}

void orc_feedforward_agc(int nsamples, float reference, int noutput_items, const orc_cf *in, orc_cf *out)
{
    orc_feedforward_agc_floor(nsamples, reference, ORC_AGC_FLOOR, noutput_items, in, out);
}

/* ------------------------------------------------------------------ */
/* msk_timing_recovery_cc  (lib/msk_timing_recovery_cc_impl.cc)        */
/* ------------------------------------------------------------------ */
struct orc_msk {
    float d_sps, d_gain, d_limit;
    orc_cf d_dly_conj_1, d_dly_conj_2, d_dly_diff_1;
    float d_mu, d_omega, d_gain_omega;
    int d_div, d_osps;
};

void orc_msk_set_sps(orc_msk *h, float sps)
{
    h->d_sps = (float)(sps / 2.0); /* :70 loop runs at 2x sps */
    h->d_omega = h->d_sps;
}
float orc_msk_get_sps(const orc_msk *h) { return h->d_sps; }
int orc_msk_set_gain(orc_msk *h, float gain)
{
    h->d_gain = gain;
    if (h->d_gain <= 0)
        return -1; /* std::out_of_range("Gain must be positive") :82 */
    h->d_gain_omega = (float)(h->d_gain * h->d_gain * 0.25);
    return 0;
}
float orc_msk_get_gain(const orc_msk *h) { return h->d_gain; }
void orc_msk_set_limit(orc_msk *h, float limit) { h->d_limit = limit; }
float orc_msk_get_limit(const orc_msk *h) { return h->d_limit; }

/* constructor :45-62 */
orc_msk *orc_msk_create(float sps, float gain, float limit, int osps, int *err)
{
    orc_msk *h = (orc_msk *)calloc(1, sizeof(*h));
    if (err)
        *err = 0;
    h->d_limit = limit;
    h->d_mu = 0.5f;
    h->d_div = 0;
    h->d_osps = osps;
    orc_msk_set_sps(h, sps);
    if (orc_msk_set_gain(h, gain) != 0) {
        if (err)
            *err = 1;
        free(h);
        return NULL;
    }
    if (osps != 1 && osps != 2) { /* :61 */
        if (err)
            *err = 2;
        free(h);
        return NULL;
    }
    return h;
}
void orc_msk_destroy(orc_msk *h) { free(h); }

/* forecast :98-105 ; ntaps() = 8 */
int orc_msk_forecast(const orc_msk *h, int noutput_items)
{
    return (int)ceil((noutput_items * h->d_sps * 2) + 3.0 * h->d_sps + (unsigned)ORC_MMSE_NTAPS);
}

void orc_msk_get_state(const orc_msk *h, float *s, int *div)
{
    s[0] = h->d_mu;
    s[1] = h->d_omega;
    s[2] = h->d_dly_conj_1.re;
    s[3] = h->d_dly_conj_1.im;
    s[4] = h->d_dly_conj_2.re;
    s[5] = h->d_dly_conj_2.im;
    s[6] = h->d_dly_diff_1.re;
    s[7] = h->d_dly_diff_1.im;
    *div = h->d_div;
}

/* Diagnostic for tools/msk_tag_stats.py (not part of the restatement): while a buffer is set,
 * every general_work call appends 4-int records -- {1, stream offset of a tag that reset the loop,
 * iterations since the previous reset, iidx before the reset - tag offset} and, at the end of the
 * call, {2, tags in range, tags used, iterations}.  Single-threaded use only. */
static int *g_msk_trace = NULL;
static int g_msk_trace_cap = 0, g_msk_trace_n = 0;
void orc_msk_set_trace(int *buf, int cap_records)
{
    g_msk_trace = buf;
    g_msk_trace_cap = cap_records;
    g_msk_trace_n = 0;
}
int orc_msk_trace_count(void) { return g_msk_trace_n; }
static void msk_trace(int a, int b, int c, int d)
{
    if (g_msk_trace && g_msk_trace_n < g_msk_trace_cap) {
        int *r = g_msk_trace + 4 * g_msk_trace_n++;
        r[0] = a, r[1] = b, r[2] = c, r[3] = d;
    }
}

/* general_work :107-206 */
int orc_msk_general_work(orc_msk *h, int noutput_items, int ninput_items, const orc_cf *in, orc_cf *out,
                         float *out2, float *out3, const orc_tag *alltags, int nalltags, uint64_t nitems_read,
                         int *consumed, int *status)
{
    int oidx = 0, iidx = 0;
    int ninp = (int)(ninput_items - 3.0 * h->d_sps); /* :119 */
    if (status)
        *status = 0;
    if (ninp <= 0) {
        *consumed = 0;
        return 0;
    }
    /* :125-130 get_tags_in_range(tags, 0, nitems_read, nitems_read+ninp, "time_est") */
    int *tq = (int *)orc_big_alloc(sizeof(int) * (nalltags > 0 ? nalltags : 1));
    int nt = 0, tpos = 0;
    for (int k = 0; k < nalltags; k++)
        if (alltags[k].key == ORC_KEY_TIME_EST && alltags[k].offset >= nitems_read &&
            alltags[k].offset < nitems_read + (uint64_t)ninp)
            tq[nt++] = k;

    orc_cf sq, dly_conj, nlin_out, in_interp;
    float err_out = 0;
    int iters = 0, last_reset = 0;
    while (oidx < noutput_items && iidx < ninp) { /* :138 */
        iters++;
        if (tpos < nt) {                            /* tags.size() > 0 */
            int offset = (int)(alltags[tq[tpos]].offset - nitems_read);
            if ((offset >= iidx) && (offset < (iidx + h->d_sps))) { /* :142 */
                float center = (float)alltags[tq[tpos]].value;
                if (center != center) { /* NaN :144-147 */
                    tpos++;
                    goto out;
                }
                if (g_msk_trace) {
                    msk_trace(1, (int)(alltags[tq[tpos]].offset), iters - last_reset, iidx - offset);
                    last_reset = iters;
                }
                h->d_mu = center;
                iidx = offset;
                if (h->d_mu < 0) {
                    h->d_mu++;
                    iidx--;
                }
                h->d_div = 0;
                h->d_omega = h->d_sps;
                h->d_dly_conj_2 = h->d_dly_conj_1;
                tpos++;
            }
        }
    out: {
        int ierr = 0;
        in_interp = orc_mmse_interpolate(&in[iidx], h->d_mu, &ierr); /* :170 */
        if (ierr && status)
            *status = 1;
    }
        sq = cmul(in_interp, in_interp);                                    /* :171 */
        dly_conj = cconj(cmul(h->d_dly_conj_2, h->d_dly_conj_2));           /* :173 */
        nlin_out = cmul(sq, dly_conj);                                      /* :174 */
        err_out = csub(nlin_out, h->d_dly_diff_1).re;                       /* :178 */
        if (h->d_div % 2) {                                                 /* :179 */
            err_out = orc_branchless_clip(err_out, 3.0f);
            h->d_omega += h->d_gain_omega * err_out;
            h->d_omega = h->d_sps + orc_branchless_clip(h->d_omega - h->d_sps, h->d_limit);
            h->d_mu += h->d_gain * err_out;
        }
        if (!(h->d_div % 2) || h->d_osps == 2) { /* :186 */
            out[oidx] = in_interp;
            if (out2)
                out2[oidx] = err_out;
            if (out3)
                out3[oidx] = h->d_mu;
            oidx++;
        }
        h->d_div++;
        h->d_dly_conj_1 = in_interp; /* :194-196 */
        h->d_dly_conj_2 = h->d_dly_conj_1;
        h->d_dly_diff_1 = nlin_out;
        h->d_mu += h->d_omega; /* :199-201 */
        iidx += (int)floor(h->d_mu);
        h->d_mu = (float)(h->d_mu - floor(h->d_mu));
    }
    if (g_msk_trace)
        msk_trace(2, nt, tpos, iters);
    orc_big_free(tq);
    *consumed = iidx; /* consume_each(iidx) */
    return oidx;
}

/* ------------------------------------------------------------------ */
/* NRZI bit tail: [GR] quadrature_demod_cf(pi/2) -> binary_slicer_fb   */
/* -> diff_decoder_bb(2) -> ais.invert (python/ais_demod.py:48-52,56;  */
/* lib/invert_impl.cc:62-64)                                           */
/* ------------------------------------------------------------------ */
void orc_bittail_init(orc_bittail *t)
{
    t->prev_sym.re = t->prev_sym.im = 0;
    t->prev_bit = 0;
}

void orc_bittail_process(orc_bittail *t, const orc_cf *syms, int n, unsigned char *bits)
{
    const float gain = (float)(M_PI / 2);
    for (int i = 0; i < n; i++) {
        /* volk_32fc_x2_multiply_conjugate_32fc generic: a * conj(b) */
        orc_cf prod = cmul(syms[i], cconj(t->prev_sym));
        float fm = gain * orc_fast_atan2f(prod.im, prod.re);
        unsigned char b = fm >= 0 ? 1 : 0;                          /* binary_slicer */
        unsigned char d = (unsigned char)(((unsigned)(b - t->prev_bit)) % 2u); /* diff_decoder_bb(2) */
        bits[i] = (d ^ 0x01) & 0x01;                                /* invert_impl.cc:63 */
        t->prev_sym = syms[i];
        t->prev_bit = b;
    }
}

/* ------------------------------------------------------------------ */
/* Template generation: digital.gmsk_mod(sps, bt) run by               */
/* digital.modulate_vector_bc(mod, data, [1])  (python/ais_demod.py:   */
/* 36-38; spec in include/ais/modulate_vector.h:48-57 and              */
/* lib/modulate_vector.cc:51-68).  [GR] gmsk_mod = packed_to_unpacked  */
/* (MSB first) -> chunks_to_symbols([-1,1]) -> interp_fir_filter_fff(  */
/* sps, convolve(firdes.gaussian(1,sps,bt,4*sps), ones(sps))) ->       */
/* frequency_modulator_fc(pi/2/sps).                                   */
/* ------------------------------------------------------------------ */
int orc_gmsk_modulate_vector(int sps, double bt, const unsigned char *data, int nbytes, orc_cf *out)
{
    int ntaps = 4 * sps;
    double *g = (double *)malloc(sizeof(double) * ntaps);
    float *gf = (float *)malloc(sizeof(float) * ntaps);
    /* firdes::gaussian(1, sps, bt, ntaps) */
    double scale = 0, dt = 1.0 / sps, s = 1.0 / (sqrt(log(2.0)) / (2 * M_PI * bt)), t0 = -0.5 * ntaps;
    for (int i = 0; i < ntaps; i++) {
        t0++;
        double ts = s * dt * t0;
        gf[i] = (float)exp(-0.5 * ts * ts);
        scale += gf[i];
    }
    for (int i = 0; i < ntaps; i++)
        gf[i] = (float)(gf[i] / scale * 1.0);
    /* numpy.convolve(gaussian_taps, (1,)*sps) -> ntaps+sps-1 taps (double in numpy) */
    int nt = ntaps + sps - 1;
    float *taps = (float *)calloc(nt, sizeof(float));
    for (int i = 0; i < nt; i++) {
        double a = 0;
        for (int k = 0; k < sps; k++)
            if (i - k >= 0 && i - k < ntaps)
                a += gf[i - k];
        taps[i] = (float)a;
    }
    int nbits = nbytes * 8;
    float *nrz = (float *)malloc(sizeof(float) * nbits);
    for (int b = 0; b < nbits; b++)
        nrz[b] = ((data[b / 8] >> (7 - (b % 8))) & 1) ? 1.0f : -1.0f;
    /* interp_fir_filter_fff(sps, taps): y[n*sps+p] = sum_k taps[k*sps+p] x[n-k] */
    float sens = (float)((M_PI / 2) / sps);
    float phase = 0;
    int o = 0;
    for (int n = 0; n < nbits; n++) {
        for (int p = 0; p < sps; p++) {
            float acc = 0;
            for (int k = 0; k * sps + p < nt; k++)
                if (n - k >= 0)
                    acc += taps[k * sps + p] * nrz[n - k];
            phase = phase + sens * acc;
            const float F_PI = (float)M_PI;
            phase = fmodf(phase + F_PI, 2.0f * F_PI) - F_PI;
            float sn, cs;
            orc_nco_sincos(phase, &sn, &cs);
            out[o].re = cs;
            out[o].im = sn;
            o++;
        }
    }
    free(g);
    free(gf);
    free(taps);
    free(nrz);
    return o;
}

/* ------------------------------------------------------------------ */
/* The chain of python/ais_demod.py:56 for ONE channel, driven by a    */
/* minimal stand-in for the GNU Radio scheduler.  One step():          */
/*   freq_sync : every complete fftlen-vector, one freqest work call   */
/*   agc       : work(n) with history 512                              */
/*   corr_est  : work(n) with history N+1 (n need not be a multiple of */
/*               the FFT filter's nsamples -- EXTENSION, see above)    */
/*   msk       : one general_work call: ninput_items = all pending     */
/*               input minus one look-ahead item, noutput_items = the  */
/*               largest count whose forecast() fits (what the         */
/*               scheduler does)                                       */
/*   bit tail  : on the produced symbols                               */
/* The HIP chain (aisx_chain_*) uses the same step contract.           */
/* ------------------------------------------------------------------ */
struct orc_demod {
    int stages, fftlen, nsym;
    orc_freqsync *fs;
    orc_cf *agc_hist; /* 511 items */
    orc_corr *corr;
    orc_cf *corr_hist; /* N items */
    uint64_t corr_written;
    orc_msk *msk;
    orc_cf *msk_buf; /* [0] = item before nitems_read, then pending items */
    int msk_pending, msk_cap;
    int msk_max_noutput; /* noutput_items the scheduler offers the timing recovery at most (0: whatever fits);
                          * gr::block::set_max_noutput_items() */
    uint64_t msk_read;
    orc_tag *store;
    int nstore, cap_store;
    orc_bittail tail;
};

#define AGC_NSAMPLES 512

orc_demod *orc_demod_create(float sps, float bits_per_sec, float gain, float limit, int fftlen, const orc_cf *symbols,
                            int nsym, int stages)
{
    orc_demod *h = (orc_demod *)calloc(1, sizeof(*h));
    int err = 0;
    h->stages = stages;
    h->fftlen = fftlen;
    h->nsym = nsym;
    /* ais_demod.py:30,34 */
    h->fs = orc_freqsync_create((double)sps * (double)bits_per_sec, bits_per_sec, fftlen);
    h->agc_hist = (orc_cf *)calloc(AGC_NSAMPLES - 1, sizeof(orc_cf));
    h->corr = orc_corr_create(symbols, nsym, sps, 1, 0.9f); /* ais_demod.py:39-42 */
    h->corr_hist = (orc_cf *)calloc(nsym, sizeof(orc_cf));
    h->msk = orc_msk_create(sps, gain, limit, 1, &err); /* ais_demod.py:43-46 */
    h->msk_cap = 1 << 16;
    h->msk_buf = (orc_cf *)orc_big_alloc(sizeof(orc_cf) * h->msk_cap);
    memset(h->msk_buf, 0, sizeof(orc_cf) * h->msk_cap);
    h->cap_store = 1024;
    h->store = (orc_tag *)malloc(sizeof(orc_tag) * h->cap_store);
    orc_bittail_init(&h->tail);
    if (!h->msk) {
        orc_demod_destroy(h);
        return NULL;
    }
    return h;
}

void orc_demod_destroy(orc_demod *h)
{
    if (!h)
        return;
    orc_freqsync_destroy(h->fs);
    free(h->agc_hist);
    orc_corr_destroy(h->corr);
    free(h->corr_hist);
    orc_msk_destroy(h->msk);
    orc_big_free(h->msk_buf);
    free(h->store);
    free(h);
}

void orc_demod_set_max_noutput(orc_demod *h, int max_noutput_items) { h->msk_max_noutput = max_noutput_items; }

int orc_demod_step(orc_demod *h, const orc_cf *in, int n, unsigned char *bits, int max_bits, orc_cf *syms_out,
                   orc_tag *tags_out, int max_tags, int *ntags_out)
{
    int nbits = 0;
    if (ntags_out)
        *ntags_out = 0;
    /* 1. freq_sync */
    orc_cf *y1 = (orc_cf *)orc_big_alloc(sizeof(orc_cf) * (size_t)(n + h->fftlen));
    int n1;
    if (h->stages & 1) {
        n1 = orc_freqsync_process(h->fs, in, n, y1, NULL);
    } else {
        memcpy(y1, in, sizeof(orc_cf) * n);
        n1 = n;
    }
    if (n1 == 0) {
        orc_big_free(y1);
        return 0;
    }
    /* 2. agc */
    orc_cf *y2 = (orc_cf *)orc_big_alloc(sizeof(orc_cf) * (size_t)n1);
    if (h->stages & 2) {
        const int H = AGC_NSAMPLES - 1;
        orc_cf *buf = (orc_cf *)orc_big_alloc(sizeof(orc_cf) * (size_t)(n1 + H));
        memcpy(buf, h->agc_hist, sizeof(orc_cf) * H);
        memcpy(buf + H, y1, sizeof(orc_cf) * n1);
        orc_feedforward_agc(AGC_NSAMPLES, 2.0f, n1, buf, y2);
        memcpy(h->agc_hist, buf + n1, sizeof(orc_cf) * H);
        orc_big_free(buf);
    } else {
        memcpy(y2, y1, sizeof(orc_cf) * n1);
    }
    /* 3. corr_est */
    const int N = h->nsym;
    orc_cf *cbuf = (orc_cf *)orc_big_alloc(sizeof(orc_cf) * (size_t)(n1 + N));
    orc_cf *y3 = (orc_cf *)orc_big_alloc(sizeof(orc_cf) * (size_t)n1);
    memcpy(cbuf, h->corr_hist, sizeof(orc_cf) * N);
    memcpy(cbuf + N, y2, sizeof(orc_cf) * n1);
    int maxt = 4 * (n1 / 1 + 1);
    orc_tag *newtags = (orc_tag *)orc_big_alloc(sizeof(orc_tag) * (size_t)maxt);
    int nnew = 0;
    orc_corr_work(h->corr, n1, cbuf, y3, NULL, h->corr_written, newtags, maxt, &nnew);
    memcpy(h->corr_hist, cbuf + n1, sizeof(orc_cf) * N);
    h->corr_written += (uint64_t)n1;
    for (int k = 0; k < nnew; k++) {
        if (tags_out && ntags_out) {
            if (*ntags_out < max_tags)
                tags_out[*ntags_out] = newtags[k];
            (*ntags_out)++;
        }
        /* insert into the tag store (multimap keyed by offset: stable by offset) */
        if (h->nstore == h->cap_store) {
            h->cap_store *= 2;
            h->store = (orc_tag *)realloc(h->store, sizeof(orc_tag) * h->cap_store);
        }
        int pos = h->nstore;
        while (pos > 0 && h->store[pos - 1].offset > newtags[k].offset) {
            h->store[pos] = h->store[pos - 1];
            pos--;
        }
        h->store[pos] = newtags[k];
        h->nstore++;
    }
    /* 4. msk timing recovery */
    if (h->msk_pending + n1 + 2 + 8 > h->msk_cap) {
        const int old_cap = h->msk_cap;
        h->msk_cap = h->msk_pending + n1 + 1024;
        orc_cf *nb = (orc_cf *)orc_big_alloc(sizeof(orc_cf) * h->msk_cap);
        memcpy(nb, h->msk_buf, sizeof(orc_cf) * old_cap);
        orc_big_free(h->msk_buf);
        h->msk_buf = nb;
    }
    memcpy(h->msk_buf + 1 + h->msk_pending, y3, sizeof(orc_cf) * n1);
    h->msk_pending += n1;
    /* the scheduler calls general_work again and again until forecast(1) no longer fits.
     * (The consumed items are dropped from the front of msk_buf once, behind the loop: with a
     * max_noutput_items there are many calls per step.) */
    /* items past the ones on offer read as zero (the reference may look a few items past
     * ninput_items when sps < 4) */
    memset(h->msk_buf + 1 + h->msk_pending, 0, sizeof(orc_cf) * 8);
    int off = 0; /* items of msk_buf consumed by the calls of this step */
    for (;;) {
        int ninput = h->msk_pending - 1; /* keep one look-ahead item out of sight */
        int nout = 0;
        if (ninput > 0) {
            nout = (int)((ninput - 3.0 * orc_msk_get_sps(h->msk) - 8) / (2.0 * orc_msk_get_sps(h->msk))) + 2;
            while (nout > 0 && orc_msk_forecast(h->msk, nout) > ninput)
                nout--;
        }
        if (h->msk_max_noutput > 0 && nout > h->msk_max_noutput)
            nout = h->msk_max_noutput;
        if (nout > max_bits - nbits)
            nout = max_bits - nbits;
        if (nout <= 0 || (int)(ninput - 3.0 * orc_msk_get_sps(h->msk)) <= 0)
            break;
        orc_cf *syms = (orc_cf *)orc_big_alloc(sizeof(orc_cf) * (size_t)nout);
        int consumed = 0, status = 0;
        int prod = orc_msk_general_work(h->msk, nout, ninput, h->msk_buf + off + 1, syms, NULL, NULL, h->store, h->nstore,
                                        h->msk_read, &consumed, &status);
        if (consumed > 0) {
            off += consumed;
            h->msk_pending -= consumed;
            h->msk_read += (uint64_t)consumed;
        }
        /* prune tags the scheduler would have dropped (offset < nitems_read); the store is sorted by offset */
        {
            int d = 0;
            while (d < h->nstore && h->store[d].offset < h->msk_read)
                d++;
            if (d > 0) {
                memmove(h->store, h->store + d, sizeof(orc_tag) * (size_t)(h->nstore - d));
                h->nstore -= d;
            }
        }
        /* 5. bit tail */
        orc_bittail_process(&h->tail, syms, prod, bits + nbits);
        if (syms_out)
            memcpy(syms_out + nbits, syms, sizeof(orc_cf) * prod);
        nbits += prod;
        orc_big_free(syms);
        /* a call that consumed nothing ends the step, whatever it produced: at sps < 4 a tag
         * with a negative centre right at nitems_read (iidx - 1, :151-154) can emit a symbol and
         * leave iidx at 0 when only one output fits; called again with the same items it would
         * do so for ever.  (A live scheduler comes back with more input instead.) */
        if (consumed <= 0)
            break;
    }
    if (off > 0)
        memmove(h->msk_buf, h->msk_buf + off, sizeof(orc_cf) * (size_t)(h->msk_pending + 1));
    orc_big_free(newtags);
    orc_big_free(cbuf);
    orc_big_free(y3);
    orc_big_free(y2);
    orc_big_free(y1);
    return nbits;
}

/* ------------------------------------------------------------------ */
/* N4: [GR] gr-digital hdlc_deframer_bp_impl::work as      */
/* python/radio.py:64 uses it, and ais.pdu_to_nmea                     */
/* (lib/pdu_to_nmea_impl.cc:63-131).                                   */
/* ------------------------------------------------------------------ */
static unsigned short orc_crc_ccitt(const unsigned char *data, int len)
{
    unsigned int POLY = 0x8408;
    unsigned short crc = 0xFFFF;
    for (int i = 0; i < len; i++) {
        crc ^= data[i];
        for (int j = 0; j < 8; j++) {
            if (crc & 0x01)
                crc = (crc >> 1) ^ POLY;
            else
                crc = (crc >> 1);
        }
    }
    return crc ^ 0xFFFF;
}

void orc_hdlc_init(orc_hdlc *h, int length_min, int length_max)
{
    memset(h, 0, sizeof(*h));
    h->length_min = length_min;
    h->length_max = length_max;
}

/* returns number of frames; frame k = out[offs[k] .. offs[k+1]) */
int orc_hdlc_work(orc_hdlc *h, const unsigned char *in, int n, unsigned char *out, int out_cap, int *offs, int max_frames)
{
    int nf = 0, used = 0;
    offs[0] = 0;
    for (int i = 0; i < n; i++) {
        unsigned char bit = in[i];
        if (h->ones >= 5) {
            if (bit) { /* six ones is a frame delimiter */
                if (h->bytectr >= h->length_min) {
                    int len = h->bytectr - 2;
                    unsigned short crc = orc_crc_ccitt(h->pktbuf, len);
                    unsigned short pktcrc = h->pktbuf[len + 1] << 8 | h->pktbuf[len];
                    if (crc == pktcrc && nf < max_frames && used + len <= out_cap) {
                        memcpy(out + used, h->pktbuf, len);
                        used += len;
                        nf++;
                        offs[nf] = used;
                    }
                    memset(h->pktbuf, 0, sizeof(h->pktbuf));
                }
                h->bitctr = 0;
                h->bytectr = 0;
            } /* else unstuff */
        } else {
            if (h->bytectr > h->length_max) {
                h->bitctr = 0;
                h->bytectr = 0;
                memset(h->pktbuf, 0, sizeof(h->pktbuf));
            } else {
                h->pktbuf[h->bytectr] >>= 1;
                if (bit)
                    h->pktbuf[h->bytectr] |= 0x80;
                h->bitctr++;
                if (h->bitctr == 8) {
                    h->bitctr = 0;
                    h->bytectr++;
                }
            }
        }
        h->ones = (bit) ? h->ones + 1 : 0;
    }
    return nf;
}

/* lib/pdu_to_nmea_impl.cc: unpack_bits :63-79, to_ascii :81-88, get_checksum :90-96,
 * to_sentence :99-124.  Returns the string length. */
int orc_pdu_to_nmea(const char *designator, const unsigned char *p, int len, char *out, int cap)
{
    int nbits = len * 8;
    int npad = (6 - (nbits % 6)) % 6;
    int nsix = (nbits + npad) / 6;
    unsigned char *up = (unsigned char *)calloc(nsix + 1, 1);
    for (int i = 0; i < nbits; i++) {
        unsigned char bit = (p[i / 8] >> (7 - (i % 8))) & 1;
        up[i / 6] |= (bit << (5 - (i % 6)));
    }
    for (int i = 0; i < npad; i++)
        up[nbits / 6] <<= 1;
    char *ascii = (char *)malloc(nsix + 1);
    for (int i = 0; i < nsix; i++) {
        char c = (char)up[i];
        if (c > 39)
            c += 8;
        c += 48;
        ascii[i] = c;
    }
    ascii[nsix] = 0;
    const int nmea_max = 56;
    int num_frags = 1 + ((nsix - 1) / nmea_max);
    int frag_id = 1, frag_offset = 0, o = 0;
    out[0] = 0;
    while (frag_id <= num_frags) {
        char sent[256];
        int flen = nsix - frag_offset < nmea_max ? nsix - frag_offset : nmea_max;
        int k = snprintf(sent, sizeof(sent), "!AIVDM,%d,%d,,%s,%.*s,%d", num_frags, frag_id, designator, flen,
                         ascii + frag_offset, npad);
        frag_id++;
        frag_offset += flen;
        unsigned char sum = 0;
        for (int i = (sent[0] == '!') ? 1 : 0; i < k; i++)
            sum ^= (unsigned char)sent[i];
        o += snprintf(out + o, cap - o, "%s%s*%02X", frag_id > 2 ? "\n" : "", sent, sum);
    }
    free(up);
    free(ascii);
    return o;
}

/* ------------------------------------------------------------------ */
/* N3: [GR] freq_xlating_fir_filter_ccf(decim, taps, center_freq, fs)  */
/* as python/radio.py:52-54 builds it per channel: band-pass taps      */
/* h[n] e^{+j 2 pi f0 n / fs}, decimating FIR (output k uses x[kD],    */
/* x[kD-1], ...; zeros before the stream start), then a rotator        */
/* e^{-j 2 pi f0 D k / fs}.  Evaluated directly in double.             */
/* ------------------------------------------------------------------ */
void orc_freq_xlating_fir(const float *taps, int ntaps, int decim, double center_freq, double fs, const orc_cf *x,
                          long nx, long k0, int nout, orc_cf *out)
{
    /* the rotated taps depend on the tap index only: formed once per call */
    double *hr = (double *)orc_big_alloc(sizeof(double) * (size_t)ntaps);
    double *hi = (double *)orc_big_alloc(sizeof(double) * (size_t)ntaps);
    for (int n = 0; n < ntaps; n++) {
        double ph = 2.0 * M_PI * center_freq * (double)n / fs;
        hr[n] = taps[n] * cos(ph);
        hi[n] = taps[n] * sin(ph);
    }
    for (int i = 0; i < nout; i++) {
        long k = k0 + i;
        double ar = 0, ai = 0;
        for (int n = 0; n < ntaps; n++) {
            long idx = k * decim - n;
            if (idx < 0 || idx >= nx)
                continue;
            ar += hr[n] * x[idx].re - hi[n] * x[idx].im;
            ai += hr[n] * x[idx].im + hi[n] * x[idx].re;
        }
        double rp = -2.0 * M_PI * center_freq * (double)decim * (double)k / fs;
        double cr = cos(rp), ci = sin(rp);
        out[i].re = (float)(ar * cr - ai * ci);
        out[i].im = (float)(ar * ci + ai * cr);
    }
    orc_big_free(hr);
    orc_big_free(hi);
}

/* [GR] firdes::low_pass(gain, fs, cutoff, transition, WIN_HAMMING) (python/radio.py:51) */
int orc_firdes_low_pass(double gain, double fs, double cutoff, double transition, float *taps, int cap)
{
    int ntaps = (int)(53.0 * fs / (22.0 * transition));
    if ((ntaps & 1) == 0)
        ntaps++;
    if (ntaps > cap)
        return -ntaps;
    int M = (ntaps - 1) / 2;
    double fwT0 = 2 * M_PI * cutoff / fs;
    for (int n = -M; n <= M; n++) {
        double w = 0.54 - 0.46 * cos((2 * M_PI * (n + M)) / (ntaps - 1));
        if (n == 0)
            taps[n + M] = (float)(fwT0 / M_PI * w);
        else
            taps[n + M] = (float)(sin(n * fwT0) / (n * M_PI) * w);
    }
    double fmax = taps[0 + M];
    for (int n = 1; n <= M; n++)
        fmax += 2 * taps[n + M];
    gain /= fmax;
    for (int i = 0; i < ntaps; i++)
        taps[i] = (float)(taps[i] * gain);
    return ntaps;
}

/* ------------------------------------------------------------------ */
/* CPU baseline B2: all cores, the  */
/* channels split across threads.  Every worker runs whole channels of */
/* nx samples through its own chain (orc_demod_create / _step /        */
/* _destroy, no shared state but the read-only FFT plans) until        */
/* budget_s seconds have passed.  Returns the channels completed;      */
/* *wall_s = the wall time they took.  Timing harness only.            */
/* ------------------------------------------------------------------ */
#include <time.h>

typedef struct {
    float sps;
    const orc_cf *symbols;
    int nsym, stages;
    const orc_cf *x;
    int nx, nsets, index;
    double budget_s, t0;
    long done;
} orc_bench_job;

static double orc_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *orc_bench_worker(void *arg)
{
    orc_bench_job *j = (orc_bench_job *)arg;
    unsigned char *bits = (unsigned char *)malloc((size_t)j->nx + 64);
    long k = 0;
    while (orc_now() - j->t0 < j->budget_s) {
        orc_demod *d = orc_demod_create(j->sps, 9600.0f, 0.04f, 0.01f, 1024, j->symbols, j->nsym, j->stages);
        if (!d)
            break;
        orc_demod_step(d, j->x + (size_t)((j->index + k) % j->nsets) * (size_t)j->nx, j->nx, bits, j->nx + 64, NULL, NULL, 0, NULL);
        orc_demod_destroy(d);
        k++;
    }
    free(bits);
    orc_pool_release();
    j->done = k;
    return NULL;
}

/* FNV-1a over the bits and the tags one channel of nx samples produces: lets a build of this
 * file with other compiler flags (the timing build, oracle/Makefile) be held to the same results */
uint64_t orc_demod_hash(float sps, const orc_cf *symbols, int nsym, int stages, const orc_cf *x, int nx)
{
    uint64_t hsh = 1469598103934665603ull;
    orc_demod *d = orc_demod_create(sps, 9600.0f, 0.04f, 0.01f, 1024, symbols, nsym, stages);
    if (!d)
        return 0;
    unsigned char *bits = (unsigned char *)malloc((size_t)nx + 64);
    const int maxt = 4 * (nx / 64 + 16);
    orc_tag *tags = (orc_tag *)malloc(sizeof(orc_tag) * (size_t)maxt);
    int nt = 0;
    const int nb = orc_demod_step(d, x, nx, bits, nx + 64, NULL, tags, maxt, &nt);
    if (nt > maxt)
        nt = maxt;
#define ORC_FNV(ptr, len)                                   \
    for (size_t q = 0; q < (size_t)(len); q++) {            \
        hsh ^= ((const unsigned char *)(ptr))[q];           \
        hsh *= 1099511628211ull;                            \
    }
    ORC_FNV(&nb, sizeof(nb));
    ORC_FNV(bits, nb > 0 ? nb : 0);
    for (int k = 0; k < nt; k++) {
        ORC_FNV(&tags[k].offset, sizeof(tags[k].offset));
        ORC_FNV(&tags[k].key, sizeof(tags[k].key));
        ORC_FNV(&tags[k].value, sizeof(tags[k].value));
    }
#undef ORC_FNV
    free(tags);
    free(bits);
    orc_demod_destroy(d);
    return hsh;
}

long orc_demod_bench_mt(int nthreads, float sps, const orc_cf *symbols, int nsym, int stages, const orc_cf *x, int nx,
                        int nsets, double budget_s, double *wall_s)
{
    if (nthreads < 1 || nsets < 1 || nx < 1)
        return 0;
    { /* create the FFT plans before the workers start */
        orc_demod *d = orc_demod_create(sps, 9600.0f, 0.04f, 0.01f, 1024, symbols, nsym, stages);
        unsigned char *bits = (unsigned char *)malloc((size_t)nx + 64);
        int warm = nx < 8192 ? nx : 8192;
        if (d) {
            orc_demod_step(d, x, warm, bits, nx + 64, NULL, NULL, 0, NULL);
            orc_demod_destroy(d);
        }
        free(bits);
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    orc_bench_job *jobs = (orc_bench_job *)calloc((size_t)nthreads, sizeof(orc_bench_job));
    const double t0 = orc_now();
    for (int i = 0; i < nthreads; i++) {
        orc_bench_job j = { sps, symbols, nsym, stages, x, nx, nsets, i, budget_s, t0, 0 };
        jobs[i] = j;
        if (pthread_create(&th[i], NULL, orc_bench_worker, &jobs[i]) != 0) {
            nthreads = i;
            break;
        }
    }
    long total = 0;
    for (int i = 0; i < nthreads; i++) {
        pthread_join(th[i], NULL);
        total += jobs[i].done;
    }
    if (wall_s)
        *wall_s = orc_now() - t0;
    free(th);
    free(jobs);
    return total;
}
