// tests/gr_mock
#pragma once
#include <pmt/pmt.h>
#include <stdint.h>
namespace gr {
struct tag_t {
    uint64_t offset;
    pmt::pmt_t key, value, srcid;
    static inline bool offset_compare(const tag_t& x, const tag_t& y) { return x.offset < y.offset; }
};
} // namespace gr
