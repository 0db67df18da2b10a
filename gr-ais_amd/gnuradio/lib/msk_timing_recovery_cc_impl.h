/* msk_timing_recovery_cc over libaisx.so: the reference's lib/msk_timing_recovery_cc_impl.h:33-69 with
 * the interpolator, the delay registers and the loop state (d_mu, d_omega, d_div ...) moved into one
 * device handle. */
#ifndef AISX_GR_AIS_MSK_TIMING_RECOVERY_CC_IMPL_H
#define AISX_GR_AIS_MSK_TIMING_RECOVERY_CC_IMPL_H

#include <ais/msk_timing_recovery_cc.h>
#include <aisx.h>

#include <vector>

namespace gr { namespace ais {

class msk_timing_recovery_cc_impl : public msk_timing_recovery_cc
{
public:
    msk_timing_recovery_cc_impl(float sps, float gain, float limit, int osps);
    ~msk_timing_recovery_cc_impl() override;

    void forecast(int noutput_items, gr_vector_int& ninput_items_required) override;
    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                     gr_vector_void_star& output_items) override;

    void set_sps(float sps) override;
    float get_sps(void) override;
    void set_gain(float gain) override;
    float get_gain(void) override;
    void set_limit(float limit) override;
    float get_limit(void) override;

private:
    aisx_msk* d_aisx;              // loop state, interpolator table, carried items and tags
    int d_osps;
    pmt::pmt_t d_time_est_key;
    std::vector<tag_t> d_found;    // get_tags_in_range's result, kept between calls
    std::vector<aisx_tag> d_tags;  // ... in the library's form
};

}} // namespace gr::ais

#endif
