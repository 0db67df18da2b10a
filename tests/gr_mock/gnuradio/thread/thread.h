// tests/gr_mock
#pragma once
#include <mutex>
namespace gr {
namespace thread {
typedef std::mutex mutex;
typedef std::unique_lock<std::mutex> scoped_lock;
} // namespace thread
} // namespace gr
