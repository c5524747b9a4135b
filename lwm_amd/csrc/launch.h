// launch.h -- HIP launch glue for the C ABI (product build).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>

namespace lwm {

inline thread_local char g_err[512] = "";

inline int fail(int code, const char* fmt, const char* a = "", long x = 0, long y = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, x, y);
    return code;
}

// memset on the stream (never synchronises)
inline int zero_device(void* p, size_t bytes, void* stream) {
    hipError_t e = hipMemsetAsync(p, 0, bytes, (hipStream_t)stream);
    if (e != hipSuccess) return fail(-3, "%s: hipMemsetAsync: %ld", "zero_device", (long)e);
    return 0;
}

// compute units of the current device (persistent kernels launch one workgroup per CU)
inline long device_cu_count() {
    static thread_local int cached_dev = -1;
    static thread_local long cached_n = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached_dev = dev;
        cached_n = n;
    }
    return cached_n;
}

template <class... KArgs, class... Args>
inline int launch(const char* name, void (*kernel)(KArgs...), long grid, int threads,
                  size_t lds_bytes, void* stream, Args... args) {
    if (grid <= 0) return 0;
    if (lds_bytes > 64 * 1024) {
        // once per kernel and host thread: the attribute is sticky, the call is not free
        static thread_local const void* raised[16];
        static thread_local int n_raised = 0;
        bool done = false;
        for (int i = 0; i < n_raised; ++i) done |= raised[i] == (const void*)kernel;
        if (!done) {
            hipError_t e = hipFuncSetAttribute((const void*)kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds_bytes);
            if (e != hipSuccess)
                return fail(-3, "%s: hipFuncSetAttribute(LDS=%ld): %ld", name, (long)lds_bytes, (long)e);
            if (n_raised < 16) raised[n_raised++] = (const void*)kernel;
        }
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3((unsigned)threads), lds_bytes,
                       (hipStream_t)stream, args...);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", name, hipGetErrorString(e));
        return -3;
    }
    return 0;
}

}  // namespace lwm
