// tests/gr_mock
#pragma once
#include <boost/shared_ptr.hpp>
namespace gnuradio {
template <class T>
boost::shared_ptr<T> get_initial_sptr(T* p)
{
    return boost::shared_ptr<T>(p);
}
} // namespace gnuradio
