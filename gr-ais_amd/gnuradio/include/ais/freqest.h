/* gr::ais::freqest -- public face of the reference's block (include/ais/freqest.h:36-49): a
 * sync_block taking vectors of fftlen complex items (fft-shifted spectra of the squared signal, as
 * python/gmsk_sync.py:23-31 wires them) and giving one float per vector, the carrier offset in Hz. */
#ifndef INCLUDED_AIS_FREQEST_H
#define INCLUDED_AIS_FREQEST_H

#include <ais/api.h>
#include <gnuradio/sync_block.h>

namespace gr {
namespace ais {

class AIS_API freqest : virtual public gr::sync_block
{
public:
    typedef boost::shared_ptr<freqest> sptr;

    static sptr make(float sample_rate, int data_rate, int fftlen);
};

} // namespace ais
} // namespace gr

#endif
