/* msk_timing_recovery_cc over libaisx.so: the reference's lib/msk_timing_recovery_cc_impl.h:33-69 with
 * the interpolator, the delay registers and the loop state (d_mu, d_omega, d_div ...) moved into one
 * device handle. */
#ifndef INCLUDED_AIS_MSK_TIMING_RECOVERY_CC_IMPL_H
#define INCLUDED_AIS_MSK_TIMING_RECOVERY_CC_IMPL_H

#include <ais/msk_timing_recovery_cc.h>
#include <aisx.h>

#include <vector>

namespace gr {
namespace ais {

class msk_timing_recovery_cc_impl : public msk_timing_recovery_cc
{
private:
    aisx_msk* d_aisx;
    int d_osps;
    pmt::pmt_t d_time_est_key;
    std::vector<tag_t> d_found;
    std::vector<aisx_tag> d_tags;

public:
    msk_timing_recovery_cc_impl(float sps, float gain, float limit, int osps);
    ~msk_timing_recovery_cc_impl();

    void forecast(int noutput_items, gr_vector_int& ninput_items_required);
    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                     gr_vector_void_star& output_items);

    void set_gain(float gain);
    float get_gain(void);
    void set_limit(float limit);
    float get_limit(void);
    void set_sps(float sps);
    float get_sps(void);
};

} // namespace ais
} // namespace gr

#endif
