// tests/gr_mock: gr::basic_block + gr::block, the members a block implementation and a scheduler touch
#pragma once
#include <gnuradio/io_signature.h>
#include <gnuradio/sptr_magic.h>
#include <gnuradio/tags.h>
#include <gnuradio/thread/thread.h>
#include <gnuradio/types.h>
#include <string>
namespace gr {

class basic_block
{
protected:
    std::string d_name;
    io_signature::sptr d_input_signature, d_output_signature;
    basic_block(void) {}
    basic_block(const std::string& name, io_signature::sptr in, io_signature::sptr out) : d_name(name), d_input_signature(in), d_output_signature(out) {}

public:
    virtual ~basic_block() {}
    std::string name() const { return d_name; }
    std::string alias() const { return d_name + "0"; }
    io_signature::sptr input_signature() const { return d_input_signature; }
    io_signature::sptr output_signature() const { return d_output_signature; }
};

class block : public basic_block
{
public:
    enum { WORK_CALLED_PRODUCE = -2, WORK_DONE = -1 };
    enum tag_propagation_policy_t { TPP_DONT = 0, TPP_ALL_TO_ALL = 1, TPP_ONE_TO_ONE = 2, TPP_CUSTOM = 3 };

    virtual ~block() {}

    unsigned history() const { return d_history; }
    void set_history(unsigned history) { d_history = history; }
    void declare_sample_delay(int which, unsigned delay)
    {
        if ((size_t)which >= mock_sample_delay.size())
            mock_sample_delay.resize((size_t)which + 1, 0);
        mock_sample_delay[(size_t)which] = delay;
    }
    void declare_sample_delay(unsigned delay) { mock_sample_delay.assign(8, delay); }
    unsigned sample_delay(int which) const { return (size_t)which < mock_sample_delay.size() ? mock_sample_delay[(size_t)which] : 0; }
    void set_output_multiple(int multiple) { d_output_multiple = multiple; }
    int output_multiple() const { return d_output_multiple; }
    void set_relative_rate(double r) { d_relative_rate = r; }
    double relative_rate() const { return d_relative_rate; }
    void enable_update_rate(bool en) { mock_update_rate = en; }
    int max_noutput_items() { return d_max_noutput_items; }
    void set_max_noutput_items(int m) { d_max_noutput_items = m, d_max_noutput_items_set = true; }
    void unset_max_noutput_items() { d_max_noutput_items_set = false; }
    bool is_set_max_noutput_items() { return d_max_noutput_items_set; }
    void set_tag_propagation_policy(tag_propagation_policy_t p) { d_tpp = p; }
    tag_propagation_policy_t tag_propagation_policy() { return d_tpp; }

    uint64_t nitems_read(unsigned which_input) { return mock_nitems_read.at(which_input); }
    uint64_t nitems_written(unsigned which_output) { return mock_nitems_written.at(which_output); }

    virtual void forecast(int noutput_items, gr_vector_int& ninput_items_required)
    {
        for (size_t i = 0; i < ninput_items_required.size(); i++)
            ninput_items_required[i] = noutput_items + (int)history() - 1;
    }
    virtual int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                             gr_vector_void_star& output_items) = 0;
    virtual bool start() { return true; }
    virtual bool stop() { return true; }

    void consume(int which_input, int how_many) { mock_consumed.at((size_t)which_input) += how_many; }
    void consume_each(int how_many)
    {
        for (size_t i = 0; i < mock_consumed.size(); i++)
            mock_consumed[i] += how_many;
    }
    void produce(int which_output, int how_many) { mock_produced.at((size_t)which_output) += how_many; }

    // ---- the test scheduler's side (not GNU Radio API)
    std::vector<uint64_t> mock_nitems_read = std::vector<uint64_t>(4, 0), mock_nitems_written = std::vector<uint64_t>(4, 0);
    std::vector<std::vector<tag_t> > mock_in_tags = std::vector<std::vector<tag_t> >(4);  // what upstream wrote onto our inputs
    std::vector<std::vector<tag_t> > mock_out_tags = std::vector<std::vector<tag_t> >(4); // what this block added
    std::vector<int> mock_consumed = std::vector<int>(4, 0), mock_produced = std::vector<int>(4, 0);
    std::vector<unsigned> mock_sample_delay;
    bool mock_update_rate = false;

protected:
    block(void) {}
    block(const std::string& name, io_signature::sptr in, io_signature::sptr out) : basic_block(name, in, out) {}

    void add_item_tag(unsigned which_output, uint64_t abs_offset, const pmt::pmt_t& key, const pmt::pmt_t& value,
                      const pmt::pmt_t& srcid = pmt::pmt_t())
    {
        tag_t t;
        t.offset = abs_offset, t.key = key, t.value = value, t.srcid = srcid;
        add_item_tag(which_output, t);
    }
    void add_item_tag(unsigned which_output, const tag_t& tag) { mock_out_tags.at(which_output).push_back(tag); }
    void get_tags_in_range(std::vector<tag_t>& v, unsigned which_input, uint64_t abs_start, uint64_t abs_end)
    {
        v.clear();
        for (const tag_t& t : mock_in_tags.at(which_input))
            if (t.offset >= abs_start && t.offset < abs_end)
                v.push_back(t);
    }
    void get_tags_in_range(std::vector<tag_t>& v, unsigned which_input, uint64_t abs_start, uint64_t abs_end, const pmt::pmt_t& key)
    {
        v.clear();
        for (const tag_t& t : mock_in_tags.at(which_input))
            if (t.offset >= abs_start && t.offset < abs_end && pmt::eqv(t.key, key))
                v.push_back(t);
    }

    gr::thread::mutex d_setlock;

private:
    unsigned d_history = 1;
    int d_output_multiple = 1;
    double d_relative_rate = 1.0;
    int d_max_noutput_items = 0;
    bool d_max_noutput_items_set = false;
    tag_propagation_policy_t d_tpp = TPP_ALL_TO_ALL;
};

typedef boost::shared_ptr<block> block_sptr;

} // namespace gr
