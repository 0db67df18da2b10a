/* corr_est_cc over libaisx.so: the reference's lib/corr_est_cc_impl.h:33-65 with the FFT filter kernel and
 * the two volk scratch buffers replaced by one device handle. */
#ifndef AISX_GR_AIS_CORR_EST_CC_IMPL_H
#define AISX_GR_AIS_CORR_EST_CC_IMPL_H

#include <ais/corr_est_cc.h>
#include <aisx.h>

#include <vector>

namespace gr { namespace ais {

class corr_est_cc_impl : public corr_est_cc
{
public:
    corr_est_cc_impl(const std::vector<gr_complex>& symbols, float sps, unsigned int mark_delay, float threshold = 0.9);
    ~corr_est_cc_impl() override;

    std::vector<gr_complex> symbols() const override;
    void set_symbols(const std::vector<gr_complex>& symbols) override;
    int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override;

private:
    void adopt_geometry();         // history / output multiple / sample delays from the handle

    pmt::pmt_t d_src_id;
    pmt::pmt_t d_keys[4];          // indexed by AISX_KEY_*: corr_start, phase_est, time_est, corr_est
    aisx_corr* d_aisx;             // taps, threshold, mark delay, filter tail and the device buffers
    std::vector<aisx_tag> d_tags;  // one work() call's tags as the library hands them back
};

}} // namespace gr::ais

#endif
