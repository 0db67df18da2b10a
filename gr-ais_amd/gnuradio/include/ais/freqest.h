/* gr::ais::freqest -- public face of the reference's block (include/ais/freqest.h:36-49): a
 * sync_block taking vectors of fftlen complex items (fft-shifted spectra of the squared signal, as
 * python/gmsk_sync.py:23-31 wires them) and giving one float per vector, the carrier offset in Hz. */
#ifndef AISX_GR_AIS_FREQEST_H
#define AISX_GR_AIS_FREQEST_H

#include <ais/api.h>
#include <gnuradio/sync_block.h>

namespace gr { namespace ais {

class AIS_API freqest : virtual public gr::sync_block
{
public:
    using sptr = boost::shared_ptr<freqest>; // (GNU Radio 3.8: boost; 3.9 and later spell it std::shared_ptr)

    static sptr make(float sample_rate, int data_rate, int fftlen);
};

}} // namespace gr::ais

#endif
