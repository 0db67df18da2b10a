/*
 * aisx.h -- C ABI of libaisx.so: the MI355X (gfx950) implementation of the gr-ais
 * per-sample demod hot path.  This is the drop-in boundary: every entry point is
 * `extern "C"`, takes plain pointers / sizes / an opaque handle, returns an int
 * status (no exceptions cross the ABI) and corresponds to one method of the
 * reference's block classes, cited below (paths under bistromath/gr-ais).
 *
 * Two ways to drive each block:
 *   *_process / *_step   batched device path: device pointers, `nchan`
 *                        independent channels laid out channel-major
 *                        (`ptr[c * stride + k]`, stride in items), work queued
 *                        on a hipStream_t passed as void* (NULL = default);
 *   *_work_host          the GNU Radio path: nchan == 1, HOST pointers exactly
 *                        as the scheduler hands them to work()/general_work();
 *                        the call stages, launches and synchronises itself.
 *
 * Per-channel carry state (correlator history, timing-loop registers, NCO
 * phase, AGC window) lives in device memory inside the handle, so a stream is
 * processed by successive calls.
 */
#ifndef AISX_H
#define AISX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AISX_VERSION 300

/* gr_complex = std::complex<float>: interleaved re, im */
typedef struct aisx_cf32 { float re, im; } aisx_cf32;

/* stream tag (gr::tag_t with a PMT double value); keys of
 * lib/corr_est_cc_impl.cc:213-256.  Tags the reference adds on output port 1
 * (:258-266) carry AISX_KEY_PORT1 or-ed into `key`. */
enum {
    AISX_KEY_CORR_START = 0,
    AISX_KEY_PHASE_EST = 1,
    AISX_KEY_TIME_EST = 2,
    AISX_KEY_CORR_EST = 3,
    AISX_KEY_PORT1 = 0x100
};
typedef struct aisx_tag {
    uint64_t offset; /* absolute item offset (nitems_written(0) + i [+ mark_delay]) */
    double value;    /* pmt::from_double payload */
    int32_t key;     /* AISX_KEY_* */
    int32_t chan;    /* channel index (0 for the GNU Radio path) */
} aisx_tag;

enum {
    AISX_OK = 0,
    AISX_ERR_INVALID = -1,      /* bad argument */
    AISX_ERR_OUT_OF_RANGE = -2, /* the reference throws std::out_of_range here */
    AISX_ERR_HIP = -3,          /* HIP runtime error, see aisx_last_error() */
    AISX_ERR_NO_DEVICE = -4,    /* no gfx950 device: there is no CPU fallback */
    AISX_ERR_OVERFLOW = -5,     /* a tag / carry buffer was too small; results truncated */
    AISX_ERR_RUNTIME = -6       /* the reference throws std::runtime_error here */
};

int aisx_version(void);
const char* aisx_last_error(void);
int aisx_device_count(int* count);
int aisx_set_device(int device);
/* measurement hook (bench.py's "copy_ceiling"): the rate, in GB/s of bytes read + bytes written,
 * a plain 16-bytes-per-lane device copy of `bytes` bytes sustains over `iters` launches, timed
 * with hipEvents and no profiler attached: what "HBM-bound" can mean on this chip next to the
 * 8 TB/s spec peak */
int aisx_util_copy_GBs(size_t bytes, int iters, float* GBs);
/* test hook: the streaming AGC kernel (k_agcw.h) forms the gain reference / max_env as a refined
 * hardware reciprocal when the reference is a power of two (the stock 2): this sweeps EVERY float
 * max_env in [2^-100, 2^100] on the device and counts those for which that differs from the
 * correctly rounded float division (*count must come back 0; *example = the largest such value) */
int aisx_util_agc_rcp_mismatches(float reference, unsigned long long* count, float* example);
/* ------------------------------------------------------------------------ */
/* corr_est_cc  (include/ais/corr_est_cc.h:85-106, lib/corr_est_cc_impl.cc)  */
/* ------------------------------------------------------------------------ */
typedef struct aisx_corr aisx_corr;

/* corr_est_cc::make(symbols, sps, mark_delay, threshold) (corr_est_cc.h:102-103,
 * ctor lib/corr_est_cc_impl.cc:48-117).  max_items bounds `n` of one call;
 * max_tags_per_chan bounds the tags one call can return per channel. */
int aisx_corr_create(aisx_corr** h, const aisx_cf32* symbols, int nsym, float sps, unsigned mark_delay,
                     float threshold, int nchan, int max_items, int max_tags_per_chan);
int aisx_corr_destroy(aisx_corr* h);
/* symbols() (corr_est_cc.h:105): d_symbols as stored (reversed conjugate) */
int aisx_corr_symbols(const aisx_corr* h, aisx_cf32* out, int cap);
/* set_symbols() (corr_est_cc.h:106, impl :132-162); keeps the reference's quirks:
 * taps stored as given (no conjugate/reverse), threshold not recomputed, the FFT
 * filter restarts from a zeroed tail.  A different length re-sizes history, output
 * multiple and mark_delay as :143-161 do (up to 2048 samples).  Must not be called
 * while a call on this handle is in flight (the reference holds d_setlock, :135). */
int aisx_corr_set_symbols(aisx_corr* h, const aisx_cf32* symbols, int nsym);
int aisx_corr_geometry(const aisx_corr* h, int* nchan, int* max_items); /* what aisx_corr_create was given */
int aisx_corr_history(const aisx_corr* h);         /* history() = nsym + 1   (:95)  */
int aisx_corr_output_multiple(const aisx_corr* h); /* fft_filter nsamples    (:84-85) */
int aisx_corr_max_noutput_items(const aisx_corr* h); /* 24*1024              (:111-112) */
float aisx_corr_threshold(const aisx_corr* h);     /* d_thresh               (:71-74) */
unsigned aisx_corr_mark_delay(const aisx_corr* h); /* d_mark_delay           (:65-66) */
uint64_t aisx_corr_nitems_written(const aisx_corr* h);
int aisx_corr_reset(aisx_corr* h); /* zero history, nitems_written = 0 */

/* One work() call of n items on every channel (lib/corr_est_cc_impl.cc:164-279).
 * d_in  : n NEW items per channel (the handle supplies the N history items);
 * d_out : n items, the input delayed by N (:184);
 * d_corr: optional port-1 output, the correlator output (:174-177, :188), or NULL.
 * Semantics of one call = one work(noutput_items = n): the peak search restarts
 * at i = 0 and the climb / centre of mass stop at the call's edge.  n need not
 * be a multiple of output_multiple().  Tags stay in device memory (feed them to
 * aisx_msk_process_stream via aisx_corr_tags_device, or fetch them with
 * aisx_corr_read_tags). */
int aisx_corr_process(aisx_corr* h, const aisx_cf32* d_in, long in_stride, aisx_cf32* d_out, long out_stride,
                      aisx_cf32* d_corr, long corr_stride, int n, void* stream);
/* Placement knob (as aisx_agc_set_lds_claim): LDS a workgroup of the F = 4096 build (templates of 513 .. 2048 items) claims
 * beyond the ~55 KB it uses.  17 408 bytes keep its workgroups off the CUs that hold a timing-recovery workgroup (55 + 17 KB
 * does not fit beside 90) while two still fit a free CU: in the 4096-channel chain the recovery then runs up to 0.25 ms shorter
 * and the step up to 3 % (5.31 against 5.49 ms on one box, nothing on another), the correlator itself a third longer (1.98
 * against 1.50 ms) -- a caller's choice
 * between the step and this kernel's own rate; aisx_chain_create leaves it alone (bench.py: config.side.corr_off_recovery_cus).
 * Results do not depend on it; default 0.  Not part of the GNU Radio API. */
int aisx_corr_set_lds_claim(aisx_corr* h, int bytes);
/* measurement hook: when on, aisx_corr_process brackets the main correlator
 * kernel with hipEvents on the launch stream; aisx_corr_last_kernel_ms waits for
 * the last bracket and returns its duration. */
int aisx_corr_set_profiling(aisx_corr* h, int on);
int aisx_corr_last_kernel_ms(aisx_corr* h, float* ms);
/* durations of the main kernel in the calls made since profiling was switched on
 * (the most recent 64 at most), oldest first */
int aisx_corr_kernel_ms_history(aisx_corr* h, float* ms, int cap, int* n);
/* device tag buffers of the last call: tags[c * cap + k], k < min(counts[c], cap).  The
 * handle rotates through three sets: the pointers of call k stay valid (for a consumer on
 * another stream) until call k+3 is launched. */
int aisx_corr_tags_device(const aisx_corr* h, const aisx_tag** d_tags, const int** d_counts, int* cap);
/* copy the last call's tags to the host, channel by channel in emission order;
 * synchronises the stream.  Returns AISX_ERR_OVERFLOW if a channel overflowed
 * max_tags_per_chan or host_cap was too small (what fits is still returned). */
int aisx_corr_read_tags(aisx_corr* h, aisx_tag* host_tags, int host_cap, int* ntags, void* stream);
/* the same for the call `back` calls before the last one (0 = the last; up to 2: the handle keeps
 * three sets) -- what a pipelined caller reads once several steps have been issued */
int aisx_corr_read_tags_back(aisx_corr* h, int back, aisx_tag* host_tags, int host_cap, int* ntags, void* stream);
/* GNU Radio path (nchan == 1): `in` = input_items[0] as the scheduler passes it
 * (history()-1 old items, then noutput_items new ones), out = output_items[0],
 * corr = output_items[1] or NULL, nitems_written = nitems_written(0). */
int aisx_corr_work_host(aisx_corr* h, const aisx_cf32* in, aisx_cf32* out, aisx_cf32* corr, int noutput_items,
                        uint64_t nitems_written, aisx_tag* tags, int tag_cap, int* ntags);

/* ------------------------------------------------------------------------ */
/* msk_timing_recovery_cc (include/ais/msk_timing_recovery_cc.h:46-69,        */
/* lib/msk_timing_recovery_cc_impl.cc) + fused NRZI bit tail                  */
/* (python/ais_demod.py:48-52, lib/invert_impl.cc:62-64)                      */
/* ------------------------------------------------------------------------ */
typedef struct aisx_msk aisx_msk;

/* make(sps, gain, limit, osps) (msk_timing_recovery_cc.h:60, ctor impl :45-62).
 * AISX_ERR_OUT_OF_RANGE if gain <= 0 or osps not in {1,2} (impl :61,:82). */
int aisx_msk_create(aisx_msk** h, float sps, float gain, float limit, int osps, int nchan, int max_items);
int aisx_msk_destroy(aisx_msk* h);
int aisx_msk_geometry(const aisx_msk* h, int* nchan, int* max_items); /* what aisx_msk_create was given */
/* what the recovery kernel's launch occupies: its workgroups (one per CU at most: each takes more than half of a CU's LDS)
 * and the LDS of one -- aisx_chain_create places the front-end kernel's workgroups by these.  Not part of the GNU Radio API. */
int aisx_msk_placement(const aisx_msk* h, int* workgroups, int* lds_bytes_per_workgroup);
int aisx_msk_set_gain(aisx_msk* h, float gain); /* :80-84, AISX_ERR_OUT_OF_RANGE if gain <= 0 */
float aisx_msk_get_gain(const aisx_msk* h);     /* :86-88 */
int aisx_msk_set_limit(aisx_msk* h, float limit); /* :90-92 */
float aisx_msk_get_limit(const aisx_msk* h);      /* :94-96 */
int aisx_msk_set_sps(aisx_msk* h, float sps);     /* :69-74 (d_sps = sps/2, omega reset) */
/* set_sps / set_limit / set_gain return AISX_ERR_INVALID for values outside what the kernel's
 * rings are sized for (sps/2 - |limit| >= 0.5, sps <= ~47, 2 (sps/2 + |limit|) + 3 |gain| <= 32);
 * after set_sps / set_limit ask aisx_msk_out_capacity() again. */
float aisx_msk_get_sps(const aisx_msk* h);        /* :76-78 returns d_sps */
int aisx_msk_forecast(const aisx_msk* h, int noutput_items); /* :98-105 */
/* items per channel the output arrays must hold: ceil((max_items + 128) / (2 (sps/2 - |limit|)))
 * outputs a call can produce without tags, plus room for the extra outputs of max_items / 64
 * time_est tags (each reset restarts the even/odd cadence, :159), times osps.  A call that would
 * need more stops there and reports AISX_MSK_ST_OUT_FULL. */
int aisx_msk_out_capacity(const aisx_msk* h);
int aisx_msk_reset(aisx_msk* h);

/* One general_work() call per channel under the stream contract:
 * the n new items are appended to the unconsumed items kept in the handle,
 * ninput_items = all of them minus one look-ahead item, noutput_items = the
 * largest count whose forecast() fits; unconsumed items and still-live
 * time_est tags are carried to the next call.  d_tags/d_tag_counts/tag_cap: this
 * call's tags as laid out by aisx_corr_tags_device (only key time_est is read,
 * impl :125-130), or NULL.  Outputs (any may be NULL): d_syms = port 0,
 * d_err = port 1, d_mu = port 2 (impl :186-191), d_bits = the unpacked NRZI
 * decoded bit per symbol; all [nchan][out_stride].  d_produced[c] = items
 * written for channel c. */
int aisx_msk_process_stream(aisx_msk* h, const aisx_cf32* d_in, long in_stride, int n, const aisx_tag* d_tags,
                            const int* d_tag_counts, int tag_cap, aisx_cf32* d_syms, float* d_err, float* d_mu,
                            uint8_t* d_bits, long out_stride, int* d_produced, void* stream);
/* The same call for a pipelined caller.  With the time-parallel recovery (k_mskp.h; osps 1, err / mu
 * ports open) the units of a call need its samples and tags but nothing of the call before: they run
 * on a stream of the handle's own, beside the previous call's join on `stream`.  `ready_event` (a
 * hipEvent_t, may be null) is what that stream waits for before it reads d_in / d_tags; without it it
 * waits for `stream` to reach this call, and nothing overlaps.  Results are identical. */
int aisx_msk_process_stream_after(aisx_msk* h, const aisx_cf32* d_in, long in_stride, int n, const aisx_tag* d_tags,
                                  const int* d_tag_counts, int tag_cap, aisx_cf32* d_syms, float* d_err, float* d_mu,
                                  uint8_t* d_bits, long out_stride, int* d_produced, void* stream, void* ready_event);
/* status word per channel of the last call, or-ed over channels (0 = clean):
 * 1 interpolator index out of range (upstream throws), 2 carry buffer overflow,
 * 4 carried-tag buffer overflow, 8 output rows full (results truncated),
 * 16 the tag list handed over was truncated by its producer (corr_est ran out of
 * max_tags_per_chan): tags are missing */
enum {
    AISX_MSK_ST_INTERP_RANGE = 1,
    AISX_MSK_ST_CARRY_OVERFLOW = 2,
    AISX_MSK_ST_TAGCARRY_OVERFLOW = 4,
    AISX_MSK_ST_OUT_FULL = 8,
    AISX_MSK_ST_TAGS_TRUNCATED = 16
};
/* gr::block::set_max_noutput_items(): under the stream contract every general_work call is offered
 * at most that many output items (0, the default: as many as the pending input allows).  With
 * GNU Radio's default buffers the scheduler never offers msk_timing_recovery_cc more than ~2000-4000;
 * a stale time_est tag blocks the later ones until the call ends (reference :140-142), so the value
 * bounds how long.  Takes effect with the next aisx_msk_process_stream. */
int aisx_msk_set_max_noutput_items(aisx_msk* h, int max_noutput_items);
/* The time-parallel recovery (gr-ais_amd/csrc/k_mskp.h), off by default.  The reference loop (:138-202) is a
 * recurrence, but two time_est tags one symbol apart reset it to a state that follows from the tags and
 * the samples alone: the loop is entered at up to `restart_points_per_channel` such pairs per call
 * (<= 64; 0 = off) by one lane each ("units"), and a join pass runs the serial loop from the carried
 * state, compares the two delay registers bit for bit at every restart point it reaches and takes over
 * the unit's symbols and end state where they agree.  Results are identical to the serial kernel's;
 * which is faster depends on the traffic.  join_kernel: 1 = the serial kernel
 * with fast-forward (default), 0 = one lane per channel, -1 = leave; max_unit_items: no unit is started
 * at a restart point further than this from the next one (0 = leave).  Applies to stream calls with
 * osps 1 and the err / mu ports open; everything else takes the serial kernel. */
int aisx_msk_set_time_parallel(aisx_msk* h, int restart_points_per_channel, int join_kernel, int max_unit_items);
int aisx_msk_get_max_noutput_items(const aisx_msk* h);
int aisx_msk_last_status(aisx_msk* h, int* status, void* stream);
/* Diagnostics of the time-parallel recovery (k_mskp.h) for the last aisx_msk_process_stream call,
 * summed over the channels: out10 (ten entries) = { restart points chosen, units whose run was taken over,
 * symbols that came from units, units that ended at the next restart point, units that ended
 * elsewhere (stale tag, end of the row), calls that took the time-parallel path, units whose end
 * state equals what the next unit assumed, out of this many, items of the longest unit, items of all
 * units }.  Waits for `stream`. */
int aisx_msk_restart_stats(aisx_msk* h, long long* out10, void* stream);
/* measurement hook, as aisx_corr_set_profiling: when on, every aisx_msk_process_stream call (serial
 * kernel) brackets the recovery kernel with hipEvents on its stream; aisx_msk_kernel_ms_history returns
 * the durations of the calls made since it was switched on (the most recent 64 at most), oldest first */
int aisx_msk_set_profiling(aisx_msk* h, int on);
int aisx_msk_kernel_ms_history(aisx_msk* h, float* ms, int cap, int* n);
/* The NRZI bit tail (quadrature demod .. invert, python/ais_demod.py:48-52) has no part in
 * the timing recurrence.  With a tail stream set (enable != 0) aisx_msk_process_stream
 * launches it there, ordered after the call's recovery kernel, so that the next call need
 * not wait for it: d_bits of a call is complete on THAT stream (aisx_msk_wait_tail makes
 * another stream wait for it); the caller must then alternate between two d_bits / d_syms /
 * d_produced buffers from call to call.  Default: off, everything on the call's stream. */
int aisx_msk_set_tail_stream(aisx_msk* h, void* tail_stream, int enable);
int aisx_msk_wait_tail(aisx_msk* h, void* stream);
/* Makes `stream` wait until the tag prepass of the last aisx_msk_process_stream call has run, i.e.
 * until that call's recovery kernel stands at the head of its queue.  The recovery kernel is 128
 * workgroups of 90 KB of LDS each: when it becomes ready at the same moment as a kernel with
 * thousands of small workgroups on another stream (both waiting for the same predecessor), those
 * fill every CU first and the recovery waits for a contiguous 90 KB until they drain (measured:
 * 1.6 ms of a 6 ms step, every other step).  A caller that pipelines the next step's sample passes
 * beside the recovery calls this on their stream right after aisx_msk_process_stream.  The first
 * call only arms the event (returns at once).  An event wait and nothing else, unless
 * aisx_msk_set_head_start has been called on the handle. */
int aisx_msk_wait_prepass(aisx_msk* h, void* stream);
/* The event of aisx_msk_wait_prepass fires for both queues at the same instant: which of them the
 * dispatcher serves first is a race (0.3 ms per 5.6 ms step when the recovery kernel loses it).  With
 * microseconds > 0, aisx_msk_wait_prepass also queues a one-wave kernel that sleeps that long on `stream`
 * behind the wait (ticks of the constant-rate wall clock, hipDeviceAttributeWallClockRate), so that the
 * recovery kernel gets there first.  Default 0: off.  aisx_chain_create switches it on for its own
 * streams (20 us; environment AISX_MSK_HEADSTART_US, 0 = off). */
int aisx_msk_set_head_start(aisx_msk* h, int microseconds);
/* GNU Radio path (nchan == 1), host pointers as general_work() receives them
 * (impl :107-206): tags = the time_est tags get_tags_in_range would return or
 * any superset, nitems_read = nitems_read(0).  *consumed is what to pass to
 * consume_each(), *produced the return value.  The reference's loop bound
 * (impl :119,:138) lets the 8-tap interpolator read in[ninput_items], one item
 * past what the scheduler announced; in_has_lookahead = 1 says that item is
 * readable (always true inside a GNU Radio circular buffer), 0 substitutes 0.
 * out_err / out_mu / out_bits may be NULL (ports not connected).  The NRZI bit tail runs only in calls
 * that pass out_bits; its state (previous symbol, previous sliced bit) carries on from the last call
 * that did, so a caller takes bits in every call or in none (the gr::ais block takes none: the tail
 * is four GNU Radio blocks of its own there, python/ais_demod.py:48-52). */
int aisx_msk_general_work_host(aisx_msk* h, int noutput_items, int ninput_items, const aisx_cf32* in, aisx_cf32* out,
                               float* out_err, float* out_mu, uint8_t* out_bits, const aisx_tag* tags, int ntags,
                               uint64_t nitems_read, int in_has_lookahead, int* consumed, int* produced);

/* ------------------------------------------------------------------------ */
/* freqest (include/ais/freqest.h:36-49, lib/freqest_impl.cc) and            */
/* square_and_fft_sync_cc (python/gmsk_sync.py:14-37)                        */
/* ------------------------------------------------------------------------ */
typedef struct aisx_freqsync aisx_freqsync;
/* square_and_fft_sync_cc(samplerate, bits_per_sec, fftlen) (gmsk_sync.py:15);
 * builds freqest::make(int(samplerate), int(bits_per_sec), fftlen) (:25). */
int aisx_freqsync_create(aisx_freqsync** h, double samplerate, double bits_per_sec, int fftlen, int nchan,
                         int max_items);
int aisx_freqsync_destroy(aisx_freqsync* h);
int aisx_freqsync_geometry(const aisx_freqsync* h, int* nchan, int* max_items, int* fftlen);
/* forget what aisx_freqsync_estimate_ahead has queued (nothing of it is committed before the pass it was
 * made for); `stream`: where the next pass will run -- it waits for what the dropped preparations still
 * have in flight */
int aisx_freqsync_drop_ahead(aisx_freqsync* h, void* stream);
/* Placement knob (as aisx_agc_set_lds_claim): LDS a one-wave workgroup of the NCO phase walk (aisx_freqsync_estimate_ahead /
 * aisx_freqsync_agc_process) claims beyond the 3 KB it uses.  aisx_chain_create sets it while the recovery leaves half of the
 * CUs free, so that no walk workgroup lands on a CU that holds a recovery workgroup -- where it also shuts the correlator's
 * workgroup out (4096 channels: the step 0.8 % shorter, the correlator 1.46 instead of 1.53 ms) -- and gives the handle its
 * previous claim back when it is destroyed.  Results do not depend on it; default 0.  Not part of the GNU Radio API. */
int aisx_freqsync_set_walk_lds_claim(aisx_freqsync* h, int bytes);
int aisx_freqsync_get_walk_lds_claim(const aisx_freqsync* h, int* bytes, int* used_bytes);
/* freqest::make(sample_rate, data_rate, fftlen) (include/ais/freqest.h:46) for the block on its own
 * (nchan == 1, aisx_freqest_work / aisx_freqest_work_host): d_offset and d_binsize from the FLOAT sample
 * rate as lib/freqest_impl.cc:46-47 compute them (aisx_freqsync_create truncates it to an int first, as
 * python/gmsk_sync.py:25 does).  max_vectors bounds noutput_items of one work() call. */
int aisx_freqest_create(aisx_freqsync** h, float sample_rate, int data_rate, int fftlen, int max_vectors);
/* ... for nchan rows of vectors, and any fftlen >= 2: freqest::work (lib/freqest_impl.cc:57-88) searches fftlen - offset
 * bins of spectra its caller transformed and needs no transform of its own.  A handle made with a vector length other
 * than 1024 serves aisx_freqest_work / aisx_freqest_work_host only: the square_and_fft_sync_cc entry points
 * (aisx_freqsync_process, _work_host, _agc_process, _estimate_ahead, aisx_chain_create) refuse it with AISX_ERR_INVALID. */
int aisx_freqest_create_n(aisx_freqsync** h, float sample_rate, int data_rate, int fftlen, int nchan, int max_vectors);
int aisx_freqsync_is_estimator_only(const aisx_freqsync* h); /* 1 for such a handle, 0 otherwise */
int aisx_freqsync_reset(aisx_freqsync* h);
/* n new items per channel; every complete fftlen-vector is processed (one
 * freqest work() call per channel); *n_out = items written per channel (a
 * multiple of fftlen); d_fhat (optional) = one estimate per vector,
 * [nchan][fhat_stride]. */
int aisx_freqsync_process(aisx_freqsync* h, const aisx_cf32* d_in, long in_stride, int n, aisx_cf32* d_out,
                          long out_stride, float* d_fhat, long fhat_stride, int* n_out, void* stream);
/* The hier block as ONE GNU Radio block (nchan == 1, HOST pointers): in = n new items, out
 * receives every complete fftlen-vector's worth of mixed items (pending ones stay in the
 * handle, as stream_to_vector keeps them), fhat (optional) one estimate per vector.  Returns the
 * items written (a multiple of fftlen) or a negative status. */
int aisx_freqsync_work_host(aisx_freqsync* h, const aisx_cf32* in, int n, aisx_cf32* out, int out_cap, float* fhat,
                            int fhat_cap);
/* freqest::work on already transformed vectors (lib/freqest_impl.cc:57-88):
 * d_vecs [nchan][nvec*fftlen] (fft-shifted spectra), d_out [nchan][nvec]. */
int aisx_freqest_work(aisx_freqsync* h, const aisx_cf32* d_vecs, long vec_stride, float* d_out, long out_stride,
                      int nvec, void* stream);
/* GNU Radio path (nchan == 1), HOST pointers exactly as the scheduler hands them to
 * freqest_impl::work (include/ais/freqest.h:36-49, lib/freqest_impl.h:39-41,
 * lib/freqest_impl.cc:57-88): in = input_items[0], noutput_items vectors of fftlen
 * gr_complex (item size 8*fftlen, :43); out = output_items[0], one float per vector.
 * maxpos is initialised once per call, as the reference's local is (:68 vs :74).
 * Returns noutput_items (:87) or a negative status. */
int aisx_freqest_work_host(aisx_freqsync* h, int noutput_items, const aisx_cf32* in, float* out);

/* ------------------------------------------------------------------------ */
/* analog.feedforward_agc_cc(nsamples, reference) (python/ais_demod.py:35)   */
/* ------------------------------------------------------------------------ */
typedef struct aisx_agc aisx_agc;
int aisx_agc_create(aisx_agc** h, int nsamples, float reference, int nchan, int max_items);
int aisx_agc_destroy(aisx_agc* h);
/* what aisx_agc_create was given; *fused_ok: the window is one aisx_freqsync_agc_process serves */
int aisx_agc_geometry(const aisx_agc* h, int* nchan, int* max_items, int* nsamples, int* fused_ok);
int aisx_agc_reset(aisx_agc* h);
/* the initial max_env of [GR] feedforward_agc_cc_impl::work: 1e-4 (default; GNU Radio 3.7/3.8
 * "float max_env = 1e-4; // avoid divide by zero, indirectly set max gain") or the 1e-12 of the
 * line upstream has commented out.  Not part of the GNU Radio API. */
int aisx_agc_set_floor(aisx_agc* h, float floor_env);
/* Which kernel serves a call is an implementation detail with one switch: calls with the stock
 * window (512) and a whole number of 512-item blocks run the streaming kernel (k_agcw.h: every
 * wave walks its own run of blocks, nothing but registers between loads and stores), all others
 * the tile kernels (k_agc.h).  Same results bit for bit; on = 0 keeps the tile kernels for every
 * call (A/B measurements, twin tests).  Default on.  Not part of the GNU Radio API. */
int aisx_agc_set_streaming(aisx_agc* h, int on);
/* Placement of the streaming kernel's workgroups when it runs BESIDE the timing recovery (the pipelined
 * chain): a workgroup uses 8 KB of LDS; claiming `bytes` more decides how many of them the dispatcher
 * puts on the 128 CUs that hold a recovery workgroup (92 160 of 163 840 B taken, 71 680 left) and on the
 * other CUs.  The recovery is a recurrence that every co-resident wave delays, and this kernel is the
 * densest arithmetic of the chain.  Measured (round 5, 4096 channels x 65536 samples, interleaved runs on
 * one box): no claim 5.65-5.86 ms per step with the recovery kernel at 5.45-5.66 ms and the correlator at
 * 2.50-2.59; 72 KB (none beside the recovery, two per free CU) 5.44-5.49 with the recovery at 5.05-5.09 and
 * the correlator at 1.89-1.94; 100 KB (one per free CU) 5.71-5.74.  aisx_chain_create sets 73 728 while the
 * recovery's workgroups (32 channels each) leave at least half of the CUs free, 49 152 otherwise (one front-end
 * workgroup beside each recovery workgroup: 8192 channels 9.40-9.42 against 9.47-9.60 ms); default 0.
 * Results do not depend on it.  Environment AISX_AGCW_LDS_PAD overrides (experiments). */
int aisx_agc_set_lds_claim(aisx_agc* h, int bytes);
/* the claim in force, and the LDS a streaming workgroup uses itself (either pointer may be NULL) */
int aisx_agc_get_lds_claim(const aisx_agc* h, int* bytes, int* used_bytes);
int aisx_agc_process(aisx_agc* h, const aisx_cf32* d_in, long in_stride, aisx_cf32* d_out, long out_stride, int n,
                     void* stream);
/* square_and_fft_sync_cc -> feedforward_agc_cc, the first two blocks of python/ais_demod.py:56, in
 * ONE pass over the samples (batched device path): the mixing with the NCO is done where the AGC
 * reads its input, the hier block's output is never stored.  Same results, bit for bit, as
 * aisx_freqsync_process followed by aisx_agc_process on its output; both handles advance as if
 * those had been called.  AGC windows that are a multiple of 8 (the stock 512). */
int aisx_freqsync_agc_process(aisx_freqsync* fs, aisx_agc* agc, const aisx_cf32* d_in, long in_stride, int n,
                              aisx_cf32* d_out, long out_stride, float* d_fhat, long fhat_stride, int* n_out, void* stream);
/* Prepares the frequency estimates (on `stream`) and the NCO phase walk (on `walk_stream`, NULL =
 * the same stream, behind the estimates) of the NEXT aisx_freqsync_agc_process call, which must
 * come with the same d_in / in_stride / n (it then waits for this preparation instead of
 * estimating itself; with other arguments the preparation is dropped).  The walk is a strict
 * recurrence per channel (one lane each, ~2 ms for 65536 samples whatever the channel count):
 * prepared one call ahead and on a stream of its own it runs beside the sample passes of the call
 * before.  A second estimate may be prepared while the first still waits for its call, provided
 * that call leaves no pending items (nothing pending now, its n a multiple of fftlen): issuing
 * estimate_ahead(call k + 1) BEFORE agc_process(call k) gives the walk all of step k to hide in
 * (what bench.py does).  Preparations are consumed in order; a call with other arguments drops all
 * of them.  d_in must stay valid and unchanged until its call. */
int aisx_freqsync_estimate_ahead(aisx_freqsync* fs, const aisx_cf32* d_in, long in_stride, int n, void* stream,
                                 void* walk_stream);
/* GNU Radio path (nchan == 1, HOST pointers): in = input_items[0] as the scheduler passes it to
 * a sync_block with set_history(nsamples): nsamples - 1 old items, then noutput_items new ones.
 * Returns noutput_items or a negative status. */
int aisx_agc_work_host(aisx_agc* h, int noutput_items, const aisx_cf32* in, aisx_cf32* out);

/* ------------------------------------------------------------------------ */
/* ais_demod (python/ais_demod.py:21-56): the demod chain as ONE pipelined    */
/* step per batch of new samples, over the stage handles above                */
/* ------------------------------------------------------------------------ */
typedef struct aisx_chain aisx_chain;
#define AISX_CHAIN_DEPTH 3 /* steps in flight: buffers and events rotate through this many sets */
/* The connect order of python/ais_demod.py:56 -- freq_sync -> agc -> (preamble_detect, 0) ->
 * clockrec -> demod -> slicer -> diff -> invert -- over handles the caller has built with that
 * file's constants (:28-47): fs = aisx_freqsync_create(sps * bits_per_sec, bits_per_sec, fftlen),
 * agc = aisx_agc_create(512, 2), corr = aisx_corr_create(template, sps, 1, 0.9), msk =
 * aisx_msk_create(sps, clockrec_gain, omega_relative_limit, 1), the last three sized for
 * max_items + fftlen items per call.  fs = agc = NULL gives the chain BASELINE.json's metric
 * names (corr_est -> msk timing recovery only).  The handles stay the caller's (tags, status,
 * setters, profiling go through them) and must outlive the chain; they must be fresh or reset
 * when the chain is created (it keeps count of the items stream_to_vector holds back), and while
 * a chain drives them they must not be called directly.  One thread at a time per chain.
 * aisx_chain_create checks the handles against its own arguments (same nchan; corr / msk / agc sized for
 * max_items + fftlen, freq_sync for max_items and the same fftlen; an AGC window the fused front end
 * serves) and returns AISX_ERR_INVALID otherwise.
 * The chain owns four streams, AISX_CHAIN_DEPTH sets of inter-stage buffers and the events that
 * order them: the sample passes of step k + 1 (one stream) run beside the timing recovery of step
 * k (a strict recurrence per channel, on its own stream), its bit tail and the NCO phase walk of
 * step k + 2 (two more).  Results are those of the stages called one after the other, bit for
 * bit.  The HIP runtime shares hardware queues between streams unless GPU_MAX_HW_QUEUES >= 8 is
 * in the environment before the first HIP call (the Python package sets it on import). */
int aisx_chain_create(aisx_chain** h, aisx_freqsync* fs, aisx_agc* agc, aisx_corr* corr, aisx_msk* msk, int nchan,
                      int max_items, int fftlen);
int aisx_chain_destroy(aisx_chain* h); /* waits for the steps in flight; the stage handles are not destroyed */
int aisx_chain_depth(void);            /* AISX_CHAIN_DEPTH */
/* One step: d_in [nchan][in_stride] holds n new items per channel, ready on `stream` (the chain's
 * streams wait for what `stream` has queued so far).  Outputs as in aisx_msk_process_stream:
 * d_syms (optional), d_bits (optional) [nchan][out_stride], d_produced [nchan]; they are complete
 * when aisx_chain_wait(step) says so, and every step in flight needs its own set (rotate through
 * AISX_CHAIN_DEPTH of them).  *step (optional) receives the step's number, counted from 0.
 * d_in_next / next_stride / n_next (optional): the input of the NEXT step, if it is already in
 * device memory: its frequency estimates and NCO phase walk are then prepared during this step
 * (aisx_freqsync_estimate_ahead).  The next call must come with exactly that pointer, stride and
 * count, and the samples must not change in between -- a live source therefore runs one buffer
 * ahead: step k is issued when block k + 1 has arrived.  Without it (NULL) every step estimates
 * for itself: same results, the phase walk (~2 ms at 65536 items) no longer hidden.
 * d_in may be reused once aisx_chain_wait_input(step) has passed (d_in_next: its own step's).
 * A step is NOT transactional: when it fails after its first stage call (a HIP error, a handle
 * misused behind the chain's back), stages already issued have advanced their histories.  The chain
 * then drops what was prepared ahead and refuses every further step (AISX_ERR_INVALID): reset the
 * stage handles and create a new chain.  Argument errors are reported before anything is issued and
 * leave the chain usable. */
int aisx_chain_step(aisx_chain* h, const aisx_cf32* d_in, long in_stride, int n, const aisx_cf32* d_in_next,
                    long next_stride, int n_next, aisx_cf32* d_syms, uint8_t* d_bits, long out_stride, int* d_produced,
                    void* stream, long long* step);
/* `stream` (host_blocks == 0) or the calling thread (host_blocks != 0) waits until the outputs of
 * `step` are complete / until its input buffer has been read for the last time */
int aisx_chain_wait(aisx_chain* h, long long step, void* stream, int host_blocks);
int aisx_chain_wait_input(aisx_chain* h, long long step, void* stream, int host_blocks);
int aisx_chain_synchronize(aisx_chain* h); /* everything issued so far has run */
/* corr_est's port-0 output of `step` (the conditioned samples delayed by the template length,
 * lib/corr_est_cc_impl.cc:184, which the timing recovery consumed): rows chan0 .. chan0 + nch - 1
 * are copied to d_dst[..][dst_stride] on `stream`, *n = items per row.  Valid for the last
 * AISX_CHAIN_DEPTH steps. */
int aisx_chain_read_corr_output(aisx_chain* h, long long step, int chan0, int nch, aisx_cf32* d_dst, long dst_stride, int* n,
                                void* stream);
/* corr_est's tags of `step` on the host (as aisx_corr_read_tags_back, but by step number: steps whose
 * front end emitted no whole vector made no corr_est call and have no tags).  Valid for the last
 * AISX_CHAIN_DEPTH steps; synchronises `stream`. */
int aisx_chain_read_tags(aisx_chain* h, long long step, aisx_tag* host_tags, int host_cap, int* ntags, void* stream);
/* the chain's streams (0 sample passes, 1 timing recovery, 2 bit tail, 3 phase walk), e.g. to
 * read a stage handle's results in order with the step that produced them */
void* aisx_chain_stream(aisx_chain* h, int which);

/* ------------------------------------------------------------------------ */
/* wideband front end (BASELINE config 5; reference analogue: one             */
/* freq_xlating_fir_filter_ccf(decim, low_pass(1, rate, 11e3, 1e3), f_off,    */
/* rate) per channel, python/radio.py:49-54): a polyphase channelizer that    */
/* computes all nlanes uniformly spaced channels f_m = m*fs/nlanes at once    */
/* ------------------------------------------------------------------------ */
typedef struct aisx_pfb aisx_pfb;
/* nlanes = 1024; decim = 1024 (critically sampled) or 512 (2x oversampled);
 * taps = the prototype low-pass (what firdes.low_pass returns), ntaps of them */
int aisx_pfb_create(aisx_pfb** h, int nlanes, int decim, const float* taps, int ntaps, int nstreams, int max_frames);
int aisx_pfb_destroy(aisx_pfb* h);
/* d_in [nstreams][n] new wideband samples (n a multiple of decim); writes
 * n/decim output items per lane: lane m of stream s is row s*nlanes + m of
 * d_out[..][out_stride] (channel-major, ready for the demod stages) */
int aisx_pfb_process(aisx_pfb* h, const aisx_cf32* d_in, long in_stride, int n, aisx_cf32* d_out, long out_stride,
                     int* nframes, void* stream);

/* ------------------------------------------------------------------------ */
/* host-side tail of the receive chain (python/radio.py:64-73): per-packet,   */
/* bytes-per-second work, plain CPU code, HOST pointers                      */
/* ------------------------------------------------------------------------ */
typedef struct aisx_hdlc aisx_hdlc;
/* digital.hdlc_deframer_bp(length_min, length_max) (python/radio.py:64): flag
 * search, bit unstuffing, bytes packed LSB first, CRC-16/X.25 check. */
int aisx_hdlc_create(aisx_hdlc** h, int length_min, int length_max);
int aisx_hdlc_destroy(aisx_hdlc* h);
/* feeds nbits unpacked bits (one per byte); every frame whose CRC checks is
 * appended to pdu_bytes, frame k = pdu_bytes[pdu_offsets[k] .. pdu_offsets[k+1]);
 * *npdus = frames found (AISX_ERR_OVERFLOW if they did not all fit). */
int aisx_hdlc_work(aisx_hdlc* h, const uint8_t* bits, int nbits, uint8_t* pdu_bytes, int pdu_cap, int* pdu_offsets,
                   int max_pdus, int* npdus);
/* ais.pdu_to_nmea(designator)::msg_to_sentence (lib/pdu_to_nmea_impl.cc:63-131):
 * writes the NUL-terminated !AIVDM sentence(s) (fragments separated by '\n');
 * returns the string length. */
int aisx_pdu_to_nmea(const char* designator, const uint8_t* pdu, int len, char* out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* AISX_H */
