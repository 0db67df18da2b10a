/*
 * ais_oracle.h -- CPU restatement (plain C) of the gr-ais demod hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke test
 * in __graft_entry__.py and bench.py's cpu_baseline leg may load it.  The
 * product (gr-ais_amd/) never includes, links or calls anything in oracle/.
 *
 * PARITY STATUS: "parity unpinned".  The reference (bistromath/gr-ais) ships
 * no tests, golden vectors or fixtures (lib/qa_ais.cc:30-36 is an empty
 * suite) and cannot be built here (GNU Radio >=3.8, VOLK, Boost, FFTW absent;
 * CMakeLists.txt:71).  This file restates the reference's control flow line
 * by line (citations on every function) plus the four third-party kernels it
 * calls (GNU Radio 3.8 fft_filter_ccc, VOLK mag^2 / dot product,
 * fast_atan2f, mmse_fir_interpolator_cc) from their published algorithms.
 * The only known answers it is pinned against are the control-flow
 * observations recorded in SURVEY.md section 4.1 (tests/test_oracle_kat.py).
 */
#ifndef AIS_ORACLE_H
#define AIS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } orc_cf;

enum { ORC_KEY_CORR_START = 0, ORC_KEY_PHASE_EST = 1, ORC_KEY_TIME_EST = 2, ORC_KEY_CORR_EST = 3 };

typedef struct {
    uint64_t offset; /* absolute item offset */
    double value;    /* pmt::from_double payload */
    int32_t key;     /* ORC_KEY_* */
    int32_t port;    /* output port the tag was added on */
} orc_tag;

/* ---- third-party kernels restated ---- */
float orc_fast_atan2f(float y, float x);
orc_cf orc_mmse_interpolate(const orc_cf *in, float mu, int *err);
float orc_branchless_clip(float x, float clip);
/* [GR] gr::fxpt (fxpt.h): float_to_fixed, sincos; and the two as frequency_modulator_fc uses them */
int32_t orc_fxpt_float_to_fixed(float x);
void orc_fxpt_float_to_fixed_n(const float *x, int32_t *out, long n);
void orc_fxpt_sincos(int32_t x, float *s, float *c);
void orc_nco_sincos(float phase, float *s, float *c);
void orc_fft(orc_cf *buf, int n, int inverse); /* in place, unnormalised */

/* ---- corr_est_cc (lib/corr_est_cc_impl.cc) ---- */
typedef struct orc_corr orc_corr;
orc_corr *orc_corr_create(const orc_cf *symbols, int nsym, float sps, unsigned mark_delay, float threshold);
void orc_corr_destroy(orc_corr *h);
int orc_corr_history(const orc_corr *h);        /* history() = N+1 */
int orc_corr_output_multiple(const orc_corr *h); /* nsamples of the FFT filter */
int orc_corr_fftsize(const orc_corr *h);
float orc_corr_threshold(const orc_corr *h);    /* d_thresh */
unsigned orc_corr_mark_delay(const orc_corr *h);
void orc_corr_taps(const orc_corr *h, orc_cf *out); /* d_symbols as stored */
void orc_corr_set_symbols(orc_corr *h, const orc_cf *symbols, int nsym);
/* one work() call: `in` has history()-1 old items then noutput new ones */
int orc_corr_work(orc_corr *h, int noutput_items, const orc_cf *in, orc_cf *out, orc_cf *corr_out /* or NULL */,
                  uint64_t nitems_written, orc_tag *tags, int max_tags, int *ntags);

/* ---- freqest (lib/freqest_impl.cc) and its wiring (python/gmsk_sync.py) ---- */
typedef struct { float binsize; int offset; int fftlen; } orc_freqest;
void orc_freqest_init(orc_freqest *f, float sample_rate, int data_rate, int fftlen);
int orc_freqest_work(const orc_freqest *f, int noutput_items, const orc_cf *in, float *out);

typedef struct orc_freqsync orc_freqsync;
orc_freqsync *orc_freqsync_create(double samplerate, double bits_per_sec, int fftlen);
void orc_freqsync_destroy(orc_freqsync *h);
/* feeds n new samples; processes every complete fftlen-vector in ONE freqest
 * work call; returns number of output samples (multiple of fftlen);
 * fhat_out (may be NULL) receives one estimate per vector. */
int orc_freqsync_process(orc_freqsync *h, const orc_cf *in, int n, orc_cf *out, float *fhat_out);

/* ---- analog.feedforward_agc_cc (3rd party, python/ais_demod.py:35) ---- */
/* `in` has nsamples-1 history items then noutput new ones */
#define ORC_AGC_FLOOR 1e-4f /* [GR] 3.7/3.8 "float max_env = 1e-4; // avoid divide by zero, indirectly set max gain" */
void orc_feedforward_agc(int nsamples, float reference, int noutput_items, const orc_cf *in, orc_cf *out);
/* the same with an explicit floor (1e-12f = the line upstream has commented out) */
void orc_feedforward_agc_floor(int nsamples, float reference, float floor_env, int noutput_items, const orc_cf *in,
                               orc_cf *out);

/* ---- msk_timing_recovery_cc (lib/msk_timing_recovery_cc_impl.cc) ---- */
typedef struct orc_msk orc_msk;
orc_msk *orc_msk_create(float sps, float gain, float limit, int osps, int *err);
void orc_msk_destroy(orc_msk *h);
int orc_msk_set_gain(orc_msk *h, float gain);
float orc_msk_get_gain(const orc_msk *h);
void orc_msk_set_limit(orc_msk *h, float limit);
float orc_msk_get_limit(const orc_msk *h);
void orc_msk_set_sps(orc_msk *h, float sps);
float orc_msk_get_sps(const orc_msk *h);
int orc_msk_forecast(const orc_msk *h, int noutput_items);
/* one general_work() call.  tags: every tag currently in the scheduler's tag
 * store for this stream (any key, any offset, sorted by offset); the function
 * applies get_tags_in_range itself.  Returns produced; *consumed = iidx. */
int orc_msk_general_work(orc_msk *h, int noutput_items, int ninput_items, const orc_cf *in, orc_cf *out,
                         float *out_err /* or NULL */, float *out_mu /* or NULL */, const orc_tag *tags,
                         int ntags, uint64_t nitems_read, int *consumed, int *status);
void orc_msk_get_state(const orc_msk *h, float *state8, int *div);
/* diagnostic (tools/msk_tag_stats.py): trace of tag resets, see ais_oracle.c */
void orc_msk_set_trace(int *buf, int cap_records);
int orc_msk_trace_count(void);

/* ---- NRZI bit tail (python/ais_demod.py:48-52 + lib/invert_impl.cc) ---- */
typedef struct { orc_cf prev_sym; unsigned char prev_bit; } orc_bittail;
void orc_bittail_init(orc_bittail *t);
void orc_bittail_process(orc_bittail *t, const orc_cf *syms, int n, unsigned char *bits);

/* ---- template generation (python/ais_demod.py:36-38) ---- */
/* gmsk_mod(sps, bt) driven by modulate_vector_bc(data, taps=[1]): returns
 * nbytes*8*sps samples */
int orc_gmsk_modulate_vector(int sps, double bt, const unsigned char *data, int nbytes, orc_cf *out);

/* ---- the chain of python/ais_demod.py:56 for one channel ---- */
typedef struct orc_demod orc_demod;
/* stages bitmask: 1 = freq_sync, 2 = agc, (corr_est and msk always on) */
orc_demod *orc_demod_create(float sps, float bits_per_sec, float gain, float limit, int fftlen, const orc_cf *symbols,
                            int nsym, int stages);
void orc_demod_destroy(orc_demod *h);
/* one chain step of n new input samples; returns number of bits produced */
/* gr::block::set_max_noutput_items() of the timing-recovery block: every general_work call is
 * offered at most that many output items (0: as many as the pending input allows) */
void orc_demod_set_max_noutput(orc_demod *h, int max_noutput_items);
int orc_demod_step(orc_demod *h, const orc_cf *in, int n, unsigned char *bits, int max_bits, orc_cf *syms_or_null,
                   orc_tag *tags_out, int max_tags, int *ntags);

/* ---- N4: hdlc_deframer_bp (python/radio.py:64) + pdu_to_nmea (lib/pdu_to_nmea_impl.cc) ---- */
typedef struct {
    int length_min, length_max, ones, bitctr, bytectr;
    unsigned char pktbuf[1024];
} orc_hdlc;
void orc_hdlc_init(orc_hdlc *h, int length_min, int length_max);
int orc_hdlc_work(orc_hdlc *h, const unsigned char *in, int n, unsigned char *out, int out_cap, int *offs, int max_frames);
int orc_pdu_to_nmea(const char *designator, const unsigned char *p, int len, char *out, int cap);

/* ---- N3: freq_xlating_fir_filter_ccf + firdes.low_pass (python/radio.py:49-54) ---- */
void orc_freq_xlating_fir(const float *taps, int ntaps, int decim, double center_freq, double fs, const orc_cf *x,
                          long nx, long k0, int nout, orc_cf *out);
int orc_firdes_low_pass(double gain, double fs, double cutoff, double transition, float *taps, int cap);

/* CPU baseline B2: `nthreads` workers each run whole channels of nx samples (taken round robin
 * from the nsets rows of x) through a chain of their own for ~budget_s seconds.  Returns the
 * channels completed, *wall_s the wall time.  Timing harness for bench.py's cpu_baseline. */
long orc_demod_bench_mt(int nthreads, float sps, const orc_cf *symbols, int nsym, int stages, const orc_cf *x, int nx,
                        int nsets, double budget_s, double *wall_s);

/* FNV-1a hash over the bits and tags one channel produces (one orc_demod_step of nx samples on a
 * fresh chain): bench.py holds the timing build of this file (other compiler flags, see Makefile)
 * to the results of the portable build with it. */
uint64_t orc_demod_hash(float sps, const orc_cf *symbols, int nsym, int stages, const orc_cf *x, int nx);
/* returns the calling thread's kept work buffers to the C library */
void orc_pool_release(void);

#ifdef __cplusplus
}
#endif
#endif
