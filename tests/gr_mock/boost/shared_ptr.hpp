// tests/gr_mock: boost::shared_ptr as GNU Radio 3.8's sptr typedefs spell it
#pragma once
#include <memory>
namespace boost {
using std::shared_ptr;
using std::dynamic_pointer_cast;
}
