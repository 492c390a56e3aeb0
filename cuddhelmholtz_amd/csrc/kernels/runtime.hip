// Memory and synchronisation entry points of the C ABI.
#include "common.hpp"

extern "C"
{
    int cuddh_hip_malloc_zeroed(void **ptr, size_t bytes)
    {
        *ptr = nullptr;
        if (bytes == 0)
            return 0;
        hipError_t e = hipMalloc(ptr, bytes);
        if (e != hipSuccess)
            return static_cast<int>(e);
        // the fill is complete when this returns: whatever stream touches the buffer next (the library's launch stream may be
        // a non-blocking one, which the null stream does not order against) sees zeros
        e = hipMemsetAsync(*ptr, 0, bytes, nullptr);
        if (e == hipSuccess)
            e = hipStreamSynchronize(nullptr);
        return static_cast<int>(e);
    }

    int cuddh_hip_free(void *ptr) { return ptr ? static_cast<int>(hipFree(ptr)) : 0; }

    int cuddh_hip_copy_h2d(void *dst, const void *h_src, size_t bytes)
    {
        return bytes ? static_cast<int>(hipMemcpy(dst, h_src, bytes, hipMemcpyHostToDevice)) : 0;
    }

    int cuddh_hip_copy_d2h(void *h_dst, const void *src, size_t bytes)
    {
        return bytes ? static_cast<int>(hipMemcpy(h_dst, src, bytes, hipMemcpyDeviceToHost)) : 0;
    }

    // Stream-ordered mirrors: the copy is queued behind the work already on `stream` (a blocking hipMemcpy on the null stream
    // is NOT ordered against a non-blocking stream) and complete when the call returns.
    int cuddh_hip_copy_h2d_on(void *dst, const void *h_src, size_t bytes, void *stream)
    {
        if (!bytes)
            return 0;
        hipStream_t st = cuddh_k::as_stream(stream);
        hipError_t e = hipMemcpyAsync(dst, h_src, bytes, hipMemcpyHostToDevice, st);
        if (e == hipSuccess)
            e = hipStreamSynchronize(st);
        return static_cast<int>(e);
    }

    int cuddh_hip_copy_d2h_on(void *h_dst, const void *src, size_t bytes, void *stream)
    {
        if (!bytes)
            return 0;
        hipStream_t st = cuddh_k::as_stream(stream);
        hipError_t e = hipMemcpyAsync(h_dst, src, bytes, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipStreamSynchronize(st);
        return static_cast<int>(e);
    }

    int cuddh_hip_copy_d2d(void *dst, const void *src, size_t bytes, void *stream)
    {
        return bytes ? static_cast<int>(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, cuddh_k::as_stream(stream))) : 0;
    }

    int cuddh_hip_memset_zero(void *ptr, size_t bytes, void *stream)
    {
        return bytes ? static_cast<int>(hipMemsetAsync(ptr, 0, bytes, cuddh_k::as_stream(stream))) : 0;
    }

    int cuddh_hip_stream_sync(void *stream) { return static_cast<int>(hipStreamSynchronize(cuddh_k::as_stream(stream))); }

    int cuddh_hip_device_sync(void) { return static_cast<int>(hipDeviceSynchronize()); }

    int cuddh_hip_device_count(void)
    {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess)
        {
            (void)hipGetLastError();
            return 0;
        }
        return n;
    }

    int cuddh_hip_host_alloc(void **ptr, size_t bytes) { return static_cast<int>(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault)); }
    int cuddh_hip_host_free(void *ptr) { return ptr ? static_cast<int>(hipHostFree(ptr)) : 0; }
    int cuddh_hip_copy_d2h_async(void *h_dst, const void *src, size_t bytes, void *stream)
    {
        return bytes ? static_cast<int>(hipMemcpyAsync(h_dst, src, bytes, hipMemcpyDeviceToHost, cuddh_k::as_stream(stream))) : 0;
    }
    int cuddh_hip_event_create(void **ev)
    {
        hipEvent_t e = nullptr;
        const hipError_t r = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        *ev = e;
        return static_cast<int>(r);
    }
    int cuddh_hip_event_record(void *ev, void *stream) { return static_cast<int>(hipEventRecord(static_cast<hipEvent_t>(ev), cuddh_k::as_stream(stream))); }
    int cuddh_hip_event_sync(void *ev) { return static_cast<int>(hipEventSynchronize(static_cast<hipEvent_t>(ev))); }
    int cuddh_hip_event_destroy(void *ev) { return ev ? static_cast<int>(hipEventDestroy(static_cast<hipEvent_t>(ev))) : 0; }

    int cuddh_hip_current_device(void)
    {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess)
        {
            (void)hipGetLastError();
            return -1;
        }
        return d;
    }

    const char *cuddh_hip_error_string(int err) { return hipGetErrorString(static_cast<hipError_t>(err)); }
}
