// Error reporting shared by host and device code.
// Behaviour contract (reference include/cuddh_error.hpp:13, source/cuddh_error.cpp:5-9):
// print a banner with the message, then assert(0).  Defined inline so device
// code in any translation unit can call it without relocatable device code.
#ifndef CUDDH_AMD_ERROR_HPP
#define CUDDH_AMD_ERROR_HPP

#include <hip/hip_runtime.h>

#include <cassert>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "cuddh_config.hpp"

namespace cuddh
{
    __host__ __device__ inline void cuddh_error(const char *msg)
    {
        printf("--- CUDDH ERROR ---\n\t%s\n-------------------\n", msg);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_trap();
#else
        // host: same observable behaviour as the reference's assert(0) for a C++ caller that does not
        // catch (the process aborts), but foreign-language hosts above the C ABI get an error instead
        throw std::runtime_error(msg);
#endif
    }

    namespace detail
    {
        /// throws std::runtime_error carrying the HIP error string when `err != 0`
        void check_hip(int err, const char *what);
    } // namespace detail
} // namespace cuddh

#endif
