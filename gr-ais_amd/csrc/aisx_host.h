// aisx_host.h -- host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/aisx.h"
#include "aisx_common.h"

namespace aisx {

static_assert(sizeof(tag_rec) == sizeof(aisx_tag), "tag layout");
static_assert(sizeof(cf) == sizeof(aisx_cf32), "complex layout");

char* err_buf(); // thread-local message buffer (aisx_lib.hip)

inline void set_err(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
}

#define AISX_HIPCHK(expr)                                                                      \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            ::aisx::set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return AISX_ERR_HIP;                                                               \
        }                                                                                      \
    } while (0)

// Experiment knobs.  The product library reads NO environment variable: its behaviour is what the API was told.
// A build made with -DAISX_EXPERIMENTS (lib/libaisx_exp.so: tools/ab_*, the profiling scripts, the tests of the
// alternative kernels) answers these look-ups from the environment; in the product build they fold to "unset".
inline const char* exp_env(const char* name)
{
#ifdef AISX_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

inline int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_err("no HIP device available (libaisx has no CPU fallback)");
        return AISX_ERR_NO_DEVICE;
    }
    return AISX_OK;
}

template <class T>
inline int dev_alloc(T** p, size_t count, bool zero = true)
{
    *p = nullptr;
    size_t bytes = sizeof(T) * (count ? count : 1);
    AISX_HIPCHK(hipMalloc((void**)p, bytes));
    if (zero)
        AISX_HIPCHK(hipMemset(*p, 0, bytes));
    return AISX_OK;
}

template <class T>
inline void dev_free(T*& p)
{
    if (p)
        (void)hipFree(p);
    p = nullptr;
}

} // namespace aisx
