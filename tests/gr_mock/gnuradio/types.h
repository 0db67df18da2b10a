// tests/gr_mock
#pragma once
#include <gnuradio/gr_complex.h>
#include <stddef.h>
#include <stdint.h>
#include <vector>
typedef std::vector<int> gr_vector_int;
typedef std::vector<unsigned int> gr_vector_uint;
typedef std::vector<void*> gr_vector_void_star;
typedef std::vector<const void*> gr_vector_const_void_star;
