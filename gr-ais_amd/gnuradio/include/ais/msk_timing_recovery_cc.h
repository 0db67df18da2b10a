/* gr::ais::msk_timing_recovery_cc -- public face of the reference's block
 * (include/ais/msk_timing_recovery_cc.h:46-69): a general block, one complex input, up to three
 * outputs (recovered symbols, timing error, mu); listens to time_est tags.  general_work() runs on
 * the MI355X through libaisx.so (lib/msk_timing_recovery_cc_impl.cc of this directory). */
#ifndef AISX_GR_AIS_MSK_TIMING_RECOVERY_CC_H
#define AISX_GR_AIS_MSK_TIMING_RECOVERY_CC_H

#include <ais/api.h>
#include <gnuradio/block.h>

namespace gr { namespace ais {

class AIS_API msk_timing_recovery_cc : virtual public gr::block
{
public:
    using sptr = boost::shared_ptr<msk_timing_recovery_cc>; // (GNU Radio 3.8: boost; 3.9 and later spell it std::shared_ptr)

    /* sps: samples per symbol; gain: loop gain (> 0); limit: relative limit of omega; osps: 1 or 2 */
    static sptr make(float sps, float gain, float limit, int osps);

    virtual void set_gain(float gain) = 0;
    virtual float get_gain(void) = 0;

    virtual void set_limit(float limit) = 0;
    virtual float get_limit(void) = 0;

    virtual void set_sps(float sps) = 0;
    virtual float get_sps(void) = 0;
};

}} // namespace gr::ais

#endif
