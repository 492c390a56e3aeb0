// Lazily mirrored host/device array.
// Contract (reference include/HostDeviceArray.hpp:25-100): move-only; nothing is
// allocated until the first read/write call; fresh allocations are zero filled on
// either side; copies are blocking; `*_write()` marks the other side stale;
// a failed device allocation throws std::runtime_error.
#ifndef CUDDH_AMD_MEMORY_HPP
#define CUDDH_AMD_MEMORY_HPP

#include <cstddef>
#include <iostream>
#include <utility>

#include "cuddh_hip.h"
#include "error.hpp"

namespace cuddh
{
    /// the stream the library launches on (set_stream, launch.hpp), as the C ABI takes it
    void *launch_stream();
}

namespace cuddh
{
    enum class MemorySpace
    {
        HOST,
        DEVICE
    };

    template <typename T>
    class HostDeviceArray
    {
    public:
        HostDeviceArray() = default;
        explicit HostDeviceArray(int n_) : n(n_) {}

        HostDeviceArray(const HostDeviceArray &) = delete;
        HostDeviceArray &operator=(const HostDeviceArray &) = delete;

        HostDeviceArray(HostDeviceArray &&o) noexcept { steal(o); }

        HostDeviceArray &operator=(HostDeviceArray &&o) noexcept
        {
            if (this != &o)
            {
                drop();
                steal(o);
            }
            return *this;
        }

        ~HostDeviceArray() { drop(); }

        int size() const { return n; }

        /// discard contents and change the length
        void resize(int new_size)
        {
            drop();
            n = new_size;
        }

        const T *read(MemorySpace m) const { return m == MemorySpace::HOST ? host_read() : device_read(); }
        T *write(MemorySpace m) { return m == MemorySpace::HOST ? host_write() : device_write(); }
        T *read_write(MemorySpace m) { return m == MemorySpace::HOST ? host_read_write() : device_read_write(); }

        const T *host_read(bool force_copy = false) const
        {
            if (n < 1)
                return nullptr;
            if (!host_fresh || force_copy)
            {
                ensure_host();
                if (dev_fresh)
                {
                    log("D -> H copy");
                    detail::check_hip(cuddh_hip_copy_d2h_on(host, dev, bytes(), launch_stream()), "HostDeviceArray device-to-host copy");
                }
            }
            host_fresh = true;
            return host;
        }

        T *host_write()
        {
            if (n < 1)
                return nullptr;
            ensure_host();
            host_fresh = true;
            dev_fresh = false;
            return host;
        }

        T *host_read_write(bool force_copy = false)
        {
            host_read(force_copy);
            return host_write();
        }

        /// hands the host buffer (allocated with new[]) to the caller, as is
        T *host_release() { return std::exchange(host, nullptr); }

        const T *device_read(bool force_copy = false) const
        {
            if (n < 1)
                return nullptr;
            if (!dev_fresh || force_copy)
            {
                ensure_device();
                if (host_fresh)
                {
                    log("H -> D copy");
                    detail::check_hip(cuddh_hip_copy_h2d_on(dev, host, bytes(), launch_stream()), "HostDeviceArray host-to-device copy");
                }
            }
            dev_fresh = true;
            return dev;
        }

        T *device_write()
        {
            if (n < 1)
                return nullptr;
            ensure_device();
            dev_fresh = true;
            host_fresh = false;
            return dev;
        }

        T *device_read_write(bool force_copy = false)
        {
            device_read(force_copy);
            return device_write();
        }

        /// hands the device buffer to the caller (free with cuddh_hip_free / hipFree)
        T *device_release() { return std::exchange(dev, nullptr); }

    private:
        std::size_t bytes() const { return static_cast<std::size_t>(n) * sizeof(T); }

        void ensure_host() const
        {
            if (!host)
            {
                log("host allocation");
                host = new T[n]();
            }
        }

        void ensure_device() const
        {
            if (!dev)
            {
                log("device allocation");
                void *p = nullptr;
                const int err = cuddh_hip_malloc_zeroed(&p, bytes());
                if (err != 0)
                    throw std::runtime_error(cuddh_hip_error_string(err));
                dev = static_cast<T *>(p);
            }
        }

        void drop()
        {
            delete[] host;
            host = nullptr;
            if (dev)
                cuddh_hip_free(dev);
            dev = nullptr;
            host_fresh = dev_fresh = false;
        }

        void steal(HostDeviceArray &o)
        {
            n = o.n;
            host = std::exchange(o.host, nullptr);
            dev = std::exchange(o.dev, nullptr);
            host_fresh = std::exchange(o.host_fresh, false);
            dev_fresh = std::exchange(o.dev_fresh, false);
        }

        void log(const char *what) const
        {
#ifdef CUDDH_LOG_MEMCPY
            std::cout << "HostDeviceArray[" << static_cast<const void *>(this) << "]: " << what << " (" << bytes() << " bytes)" << std::endl;
#else
            (void)what;
#endif
        }

        int n = 0;
        mutable T *host = nullptr;
        mutable T *dev = nullptr;
        mutable bool host_fresh = false;
        mutable bool dev_fresh = false;
    };

    typedef HostDeviceArray<double> host_device_dvec;
    typedef HostDeviceArray<int> host_device_ivec;
} // namespace cuddh

#endif
