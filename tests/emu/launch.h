// tests/emu/launch.h -- host-emulation launch glue (TEST INFRASTRUCTURE ONLY).
#pragma once
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

namespace lwm {

inline thread_local char g_err[512] = "";

inline int fail(int code, const char* fmt, const char* a = "", long x = 0, long y = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, x, y);
    return code;
}

inline int zero_device(void* p, size_t bytes, void* stream) {
    (void)stream;
    memset(p, 0, bytes);
    return 0;
}
// a few "XCDs" worth of persistent workgroups; LWM_EMU_CUS: a test of a split that depends on the device's size
inline long device_cu_count() {
    const char* e = getenv("LWM_EMU_CUS");
    const long n = e ? atol(e) : 0;
    return n > 0 ? n : 24;
}

template <class... KArgs, class... Args>
inline int launch(const char* name, void (*kernel)(KArgs...), long grid, int threads,
                  size_t lds_bytes, void* stream, Args... args) {
    (void)stream;
    if (getenv("LWM_EMU_TRACE")) fprintf(stderr, "emu-launch %s grid=%ld threads=%d\n", name, grid, threads);
    if (grid <= 0) return 0;
    emu::launch(emu::Dim3{(int)grid, 1, 1}, threads, lds_bytes ? lds_bytes : 16,
                [=]() { kernel(args...); });
    return 0;
}

}  // namespace lwm
