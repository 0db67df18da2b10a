/* gr::ais::msk_timing_recovery_cc, general_work() on the MI355X.
 *
 * Takes the place of the reference's lib/msk_timing_recovery_cc_impl.cc.  The loop of :107-206 -- the
 * time_est resets, the 8-tap MMSE interpolation, the Gardner-style error on the squared signal, the
 * omega / mu update with its clipping -- runs inside libaisx.so (aisx_msk_*, include/aisx.h) with its
 * state carried from call to call in the handle; this file hands it the scheduler's buffers and the
 * time_est tags of the window, and reports back what was consumed and produced. */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif

#include "msk_timing_recovery_cc_impl.h"

#include <gnuradio/io_signature.h>

#include <stdexcept>

namespace gr { namespace ais {

msk_timing_recovery_cc::sptr msk_timing_recovery_cc::make(float sps, float gain, float limit, int osps)
{
    return gnuradio::get_initial_sptr(new msk_timing_recovery_cc_impl(sps, gain, limit, osps));
}

static_assert(sizeof(gr_complex) == sizeof(aisx_cf32), "gr_complex must be two packed floats");

// status of a setter -> the exception the reference's setter throws (:80-84), or the library's own
static void raise_unless_ok(int rc, const char* what)
{
    if (rc == AISX_OK)
        return;
    if (rc == AISX_ERR_OUT_OF_RANGE)
        throw std::out_of_range(std::string(what) + ": " + aisx_last_error());
    throw std::runtime_error(std::string(what) + ": " + aisx_last_error());
}

msk_timing_recovery_cc_impl::msk_timing_recovery_cc_impl(float sps, float gain, float limit, int osps)
    : gr::block("msk_timing_recovery_cc", gr::io_signature::make(1, 1, sizeof(gr_complex)),
                gr::io_signature::make3(1, 3, sizeof(gr_complex), sizeof(float), sizeof(float))),
      d_aisx(nullptr),
      d_osps(osps),
      d_time_est_key(pmt::intern("time_est"))
{
    // gain <= 0 and osps outside {1, 2}: std::out_of_range as in the reference (:61, :82).  One channel;
    // the scheduler's buffers for 8-byte items hold 8192 of them by default, 64 K leaves room for
    // set_min_output_buffer() users.
    raise_unless_ok(aisx_msk_create(&d_aisx, sps, gain, limit, osps, /*nchan*/ 1, /*max_items*/ 1 << 16), "msk_timing_recovery_cc");
    set_relative_rate((double)osps / (double)sps); // :72
    enable_update_rate(true);                      // :59: tags downstream follow the measured rate
}

msk_timing_recovery_cc_impl::~msk_timing_recovery_cc_impl() { aisx_msk_destroy(d_aisx); }

void msk_timing_recovery_cc_impl::set_sps(float sps)
{
    raise_unless_ok(aisx_msk_set_sps(d_aisx, sps), "msk_timing_recovery_cc::set_sps"); // d_sps = sps / 2, omega restarts (:69-74)
    set_relative_rate((double)d_osps / (double)sps);
}
float msk_timing_recovery_cc_impl::get_sps(void) { return aisx_msk_get_sps(d_aisx); } // the halved value, as :76-78

void msk_timing_recovery_cc_impl::set_gain(float gain) { raise_unless_ok(aisx_msk_set_gain(d_aisx, gain), "msk_timing_recovery_cc::set_gain"); }
float msk_timing_recovery_cc_impl::get_gain(void) { return aisx_msk_get_gain(d_aisx); }

void msk_timing_recovery_cc_impl::set_limit(float limit) { raise_unless_ok(aisx_msk_set_limit(d_aisx, limit), "msk_timing_recovery_cc::set_limit"); }
float msk_timing_recovery_cc_impl::get_limit(void) { return aisx_msk_get_limit(d_aisx); }

void msk_timing_recovery_cc_impl::forecast(int noutput_items, gr_vector_int& ninput_items_required)
{
    const int need = aisx_msk_forecast(d_aisx, noutput_items); // :98-105
    for (size_t i = 0; i < ninput_items_required.size(); i++)
        ninput_items_required[i] = need;
}

int msk_timing_recovery_cc_impl::general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                                              gr_vector_void_star& output_items)
{
    // the time_est tags of the window the scheduler offers (:125-130)
    const uint64_t nread = nitems_read(0);
    d_found.clear();
    get_tags_in_range(d_found, 0, nread, nread + (uint64_t)ninput_items[0], d_time_est_key);
    d_tags.resize(d_found.size());
    for (size_t k = 0; k < d_found.size(); k++) {
        d_tags[k].offset = d_found[k].offset;
        d_tags[k].value = pmt::to_double(d_found[k].value);
        d_tags[k].key = AISX_KEY_TIME_EST;
        d_tags[k].chan = 0;
    }

    int consumed = 0, produced = 0;
    const int rc = aisx_msk_general_work_host(
        d_aisx, noutput_items, ninput_items[0], static_cast<const aisx_cf32*>(input_items[0]), static_cast<aisx_cf32*>(output_items[0]),
        output_items.size() >= 2 ? static_cast<float*>(output_items[1]) : nullptr, // the error signal (:187-190)
        output_items.size() >= 3 ? static_cast<float*>(output_items[2]) : nullptr, // mu
        /*out_bits*/ nullptr, d_tags.data(), (int)d_tags.size(), nread,
        /*in_has_lookahead: the reference's loop reads in[ninput_items] (:119, :138); inside a GNU Radio
           circular buffer that item is mapped memory, and so it is here*/ 1,
        &consumed, &produced);
    if (rc != AISX_OK) // AISX_ERR_RUNTIME: the interpolator's "imu out of bounds" (mmse_fir_interpolator_cc)
        throw std::runtime_error(std::string("msk_timing_recovery_cc::general_work: ") + aisx_last_error());

    consume_each(consumed); // :204
    return produced;        // :205
}

}} // namespace gr::ais
