/* gr::ais::freqest, work() on the MI355X.
 *
 * Takes the place of the reference's lib/freqest_impl.cc: per input vector (an fft-shifted spectrum of
 * the squared signal) the two-line search of :57-88 -- the bin pair `d_offset` apart with the largest
 * summed magnitude -- and the conversion of its position to Hz run inside libaisx.so
 * (aisx_freqest_work_host, include/aisx.h).  A flowgraph that wants the whole of python/gmsk_sync.py on
 * the device in one go uses aisx_freqsync_work_host instead. */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif

#include "freqest_impl.h"

#include <gnuradio/io_signature.h>

#include <stdexcept>

namespace gr { namespace ais {

freqest::sptr freqest::make(float sample_rate, int data_rate, int fftlen)
{
    return gnuradio::get_initial_sptr(new freqest_impl(sample_rate, data_rate, fftlen));
}

static_assert(sizeof(gr_complex) == sizeof(aisx_cf32), "gr_complex must be two packed floats");

freqest_impl::freqest_impl(float sample_rate, int data_rate, int fftlen)
    : gr::sync_block("freqest", gr::io_signature::make(1, 1, sizeof(gr_complex) * fftlen), gr::io_signature::make(1, 1, sizeof(float))),
      d_aisx(nullptr)
{
    // (64 vectors of staging to begin with; a longer work() call grows it)
    if (aisx_freqest_create(&d_aisx, sample_rate, data_rate, fftlen, 64) != AISX_OK)
        throw std::runtime_error(std::string("freqest: ") + aisx_last_error());
}

freqest_impl::~freqest_impl() { aisx_freqsync_destroy(d_aisx); }

int freqest_impl::work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items)
{
    const int rc = aisx_freqest_work_host(d_aisx, noutput_items, static_cast<const aisx_cf32*>(input_items[0]), static_cast<float*>(output_items[0]));
    if (rc < 0)
        throw std::runtime_error(std::string("freqest::work: ") + aisx_last_error());
    return rc; // = noutput_items (:87)
}

}} // namespace gr::ais
