// tests/gr_mock
#pragma once
#include <boost/shared_ptr.hpp>
#include <vector>
namespace gr {
class io_signature
{
    int d_min, d_max;
    std::vector<int> d_sizes;
    io_signature(int mn, int mx, const std::vector<int>& s) : d_min(mn), d_max(mx), d_sizes(s) {}

public:
    typedef boost::shared_ptr<io_signature> sptr;
    static const int IO_INFINITE = -1;
    static sptr make(int mn, int mx, int size) { return sptr(new io_signature(mn, mx, std::vector<int>(1, size))); }
    static sptr make2(int mn, int mx, int s1, int s2)
    {
        std::vector<int> v;
        v.push_back(s1), v.push_back(s2);
        return sptr(new io_signature(mn, mx, v));
    }
    static sptr make3(int mn, int mx, int s1, int s2, int s3)
    {
        std::vector<int> v;
        v.push_back(s1), v.push_back(s2), v.push_back(s3);
        return sptr(new io_signature(mn, mx, v));
    }
    int min_streams() const { return d_min; }
    int max_streams() const { return d_max; }
    int sizeof_stream_item(int i) const { return d_sizes[(size_t)i < d_sizes.size() ? (size_t)i : d_sizes.size() - 1]; }
};
} // namespace gr
