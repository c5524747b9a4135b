// tests/emu/launch.h -- host-emulation launch glue (TEST INFRASTRUCTURE ONLY).
#pragma once
#include <stdio.h>

namespace lwm {

inline thread_local char g_err[512] = "";

inline int fail(int code, const char* fmt, const char* a = "", long x = 0, long y = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, x, y);
    return code;
}

template <class... KArgs, class... Args>
inline int launch(const char* name, void (*kernel)(KArgs...), long grid, int threads,
                  size_t lds_bytes, void* stream, Args... args) {
    (void)name; (void)stream;
    if (grid <= 0) return 0;
    emu::launch(emu::Dim3{(int)grid, 1, 1}, threads, lds_bytes ? lds_bytes : 16,
                [=]() { kernel(args...); });
    return 0;
}

}  // namespace lwm
