// Library-wide runtime state: the stream launches go to, and HIP error checking.
#include <stdexcept>
#include <string>

#include "cuddh/error.hpp"
#include "cuddh/launch.hpp"
#include "cuddh_hip.h"

namespace cuddh
{
    namespace
    {
        // per host thread: a thread that never calls set_stream launches on the null stream, and two threads driving the
        // library on their own streams do not race on this variable (the reference has a single global stream: the null one)
        thread_local hipStream_t g_stream = nullptr;
    }

    hipStream_t stream() { return g_stream; }
    void set_stream(hipStream_t s) { g_stream = s; }
    void *launch_stream() { return g_stream; }

    namespace detail
    {
        void check_hip(int err, const char *what)
        {
            if (err != 0)
                throw std::runtime_error(std::string(what) + ": " + cuddh_hip_error_string(err));
        }
    } // namespace detail
} // namespace cuddh
