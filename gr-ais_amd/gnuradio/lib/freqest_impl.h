/* freqest over libaisx.so (the reference's lib/freqest_impl.h:28-45; d_binsize and d_offset live in the handle). */
#ifndef INCLUDED_AIS_FREQEST_IMPL_H
#define INCLUDED_AIS_FREQEST_IMPL_H

#include <ais/freqest.h>
#include <aisx.h>

namespace gr {
namespace ais {

class freqest_impl : public freqest
{
private:
    aisx_freqsync* d_aisx;

public:
    freqest_impl(float sample_rate, int data_rate, int fftlen);
    ~freqest_impl();

    int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items);
};

} // namespace ais
} // namespace gr

#endif
