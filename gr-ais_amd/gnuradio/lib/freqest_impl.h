/* freqest over libaisx.so (the reference's lib/freqest_impl.h:28-45; d_binsize and d_offset live in the handle). */
#ifndef AISX_GR_AIS_FREQEST_IMPL_H
#define AISX_GR_AIS_FREQEST_IMPL_H

#include <ais/freqest.h>
#include <aisx.h>

namespace gr { namespace ais {

class freqest_impl : public freqest
{
public:
    freqest_impl(float sample_rate, int data_rate, int fftlen);
    ~freqest_impl() override;
    int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override;

private:
    aisx_freqsync* d_aisx; // offset between the two spectral lines, bin width, staging buffers: in the handle
};

}} // namespace gr::ais

#endif
