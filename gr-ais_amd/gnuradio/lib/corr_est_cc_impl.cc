/* gr::ais::corr_est_cc, work() on the MI355X.
 *
 * Takes the place of the reference's lib/corr_est_cc_impl.cc.  Everything that file computes -- the
 * reversed conjugate taps and the threshold (:58-74), the FFT filter with its tail (:77-85, :188), the
 * |corr|^2 peak search with its climb and centre of mass (:190-256), the seven tags per hit (:213-266) --
 * happens inside libaisx.so (aisx_corr_*, include/aisx.h); this file is the GNU Radio side only: block
 * geometry, the scheduler's buffers in and out, tags onto the stream. */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif

#include "corr_est_cc_impl.h"

#include <gnuradio/io_signature.h>

#include <stdexcept>

namespace gr { namespace ais {

corr_est_cc::sptr corr_est_cc::make(const std::vector<gr_complex>& symbols, float sps, unsigned int mark_delay, float threshold)
{
    return gnuradio::get_initial_sptr(new corr_est_cc_impl(symbols, sps, mark_delay, threshold));
}

static_assert(sizeof(gr_complex) == sizeof(aisx_cf32), "gr_complex must be two packed floats");

corr_est_cc_impl::corr_est_cc_impl(const std::vector<gr_complex>& symbols, float sps, unsigned int mark_delay, float threshold)
    : sync_block("corr_est_cc", io_signature::make(1, 1, sizeof(gr_complex)), io_signature::make(1, 2, sizeof(gr_complex))),
      d_src_id(pmt::intern(alias())),
      d_aisx(nullptr)
{
    d_keys[AISX_KEY_CORR_START] = pmt::intern("corr_start");
    d_keys[AISX_KEY_PHASE_EST] = pmt::intern("phase_est");
    d_keys[AISX_KEY_TIME_EST] = pmt::intern("time_est");
    d_keys[AISX_KEY_CORR_EST] = pmt::intern("corr_est");

    // one channel; a work() call never exceeds max_noutput_items (:111-112); at most one hit every
    // other sample (a peak needs a sample below it on either side), seven tags per hit
    const int nitems = 24 * 1024;
    const int rc = aisx_corr_create(&d_aisx, reinterpret_cast<const aisx_cf32*>(symbols.data()), (int)symbols.size(), sps, mark_delay,
                                    threshold, /*nchan*/ 1, nitems, 7 * (nitems / 2 + 2));
    if (rc != AISX_OK)
        throw std::runtime_error(std::string("corr_est_cc: ") + aisx_last_error());
    d_tags.resize(7 * (nitems / 2 + 2));
    adopt_geometry();
    set_max_noutput_items(aisx_corr_max_noutput_items(d_aisx));
}

corr_est_cc_impl::~corr_est_cc_impl() { aisx_corr_destroy(d_aisx); }

void corr_est_cc_impl::adopt_geometry()
{
    const int nsym = aisx_corr_history(d_aisx) - 1;
    set_output_multiple(aisx_corr_output_multiple(d_aisx)); // the FFT filter's block length (:84-85, :143-145)
    set_history(nsym + 1);                                   // the delay line the tags are placed back into (:95, :155)
    declare_sample_delay(1, 0);                              // :97, :157
    declare_sample_delay(0, nsym);                           // :98, :158
}

std::vector<gr_complex> corr_est_cc_impl::symbols() const
{
    // what the reference keeps in d_symbols: the reversed conjugate after make(), the vector as given
    // after set_symbols() (:124-128)
    std::vector<gr_complex> s((size_t)aisx_corr_history(d_aisx) - 1);
    aisx_corr_symbols(d_aisx, reinterpret_cast<aisx_cf32*>(s.data()), (int)s.size());
    return s;
}

void corr_est_cc_impl::set_symbols(const std::vector<gr_complex>& symbols)
{
    gr::thread::scoped_lock lock(d_setlock); // work() holds it too (:135, :169): no call is in flight
    const int rc = aisx_corr_set_symbols(d_aisx, reinterpret_cast<const aisx_cf32*>(symbols.data()), (int)symbols.size());
    if (rc != AISX_OK)
        throw std::runtime_error(std::string("corr_est_cc::set_symbols: ") + aisx_last_error());
    adopt_geometry();
}

int corr_est_cc_impl::work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items)
{
    gr::thread::scoped_lock lock(d_setlock);

    const aisx_cf32* in = static_cast<const aisx_cf32*>(input_items[0]); // history()-1 old items, then the new ones
    aisx_cf32* out = static_cast<aisx_cf32*>(output_items[0]);
    aisx_cf32* corr = output_items.size() > 1 ? static_cast<aisx_cf32*>(output_items[1]) : nullptr;

    int ntags = 0;
    const int rc = aisx_corr_work_host(d_aisx, in, out, corr, noutput_items, nitems_written(0), d_tags.data(), (int)d_tags.size(), &ntags);
    if (rc != AISX_OK)
        throw std::runtime_error(std::string("corr_est_cc::work: ") + aisx_last_error());

    // the library returns the tags in the order the reference adds them (:213-266); port 1's copies
    // (when that output is connected) carry AISX_KEY_PORT1
    for (int k = 0; k < ntags; k++) {
        const aisx_tag& t = d_tags[(size_t)k];
        add_item_tag((t.key & AISX_KEY_PORT1) ? 1 : 0, t.offset, d_keys[t.key & 3], pmt::from_double(t.value), d_src_id);
    }
    return noutput_items;
}

}} // namespace gr::ais
