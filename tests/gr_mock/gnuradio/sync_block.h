// tests/gr_mock: gr::sync_block -- work() in place of general_work(), as many items out as in
#pragma once
#include <gnuradio/block.h>
namespace gr {
class sync_block : public block
{
protected:
    sync_block(void) {} // lets pure interface classes derive virtually
    sync_block(const std::string& name, io_signature::sptr in, io_signature::sptr out) : block(name, in, out) {}

public:
    virtual int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) = 0;
    void forecast(int noutput_items, gr_vector_int& ninput_items_required)
    {
        for (size_t i = 0; i < ninput_items_required.size(); i++)
            ninput_items_required[i] = noutput_items + (int)history() - 1;
    }
    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items)
    {
        (void)ninput_items;
        const int r = work(noutput_items, input_items, output_items);
        if (r > 0)
            consume_each(r);
        return r;
    }
};
} // namespace gr
