/* gr::ais::corr_est_cc -- the public face is the reference's (include/ais/corr_est_cc.h:94-107): a
 * sync_block, one complex input, one or two complex outputs (port 0 = the input delayed by the length
 * of the sync word and tagged corr_start / phase_est / time_est / corr_est, port 1 = the correlator
 * output).  Flowgraphs, GRC files and python/ais_demod.py:39-42 keep working unchanged; what changes
 * is lib/corr_est_cc_impl.cc, whose work() runs on the MI355X through libaisx.so. */
#ifndef AISX_GR_AIS_CORR_EST_CC_H
#define AISX_GR_AIS_CORR_EST_CC_H

#include <ais/api.h>
#include <gnuradio/sync_block.h>

#include <vector>

namespace gr { namespace ais {

class AIS_API corr_est_cc : virtual public sync_block
{
public:
    using sptr = boost::shared_ptr<corr_est_cc>; // (GNU Radio 3.8: boost; 3.9 and later spell it std::shared_ptr)

    /* symbols: the sync word at `sps` samples per symbol; mark_delay: where on the correlation peak the
     * tags go; threshold: fraction of the sync word's autocorrelation peak (squared) that counts as a hit */
    static sptr make(const std::vector<gr_complex>& symbols, float sps, unsigned int mark_delay, float threshold = 0.9);

    virtual std::vector<gr_complex> symbols() const = 0;
    virtual void set_symbols(const std::vector<gr_complex>& symbols) = 0;
};

}} // namespace gr::ais

#endif
