// conv_prof.cpp -- phase timestamps of the patch convolution kernels (a -DLWM_CONV_PROF-style variant library built from
// profiles/r06_conv_prof.patch: every workgroup stamps s_memtime at entry / patch landed / main loop done / stores issued /
// stores acknowledged, plus HW_ID and XCC_ID):   conv_prof <variant .so> <out dir> [frames=16]
// Writes <out dir>/<shape>.bin = [grid][8] uint64; scripts/conv_prof_report.py turns them into per-CU timelines.
// Build: hipcc -O2 --offload-arch=gfx950 -I include -o scripts/micro/conv_prof scripts/micro/conv_prof.cpp -ldl
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "lwm_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__global__ void fill_f32(float* p, size_t n, uint32_t seed, float amp) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = ((h & 0xffffff) * (1.0f / 8388608.0f) - 1.0f) * amp;
    }
}
struct Shape { const char* name; int H, W, Cin, Cout, up, res; };
int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const int B = argc > 3 ? atoi(argv[3]) : 16;
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    auto conv = (int (*)(const LwmConvArgs*, void*))dlsym(lib, "lwm_conv2d_nhwc_f32");
    auto prof_set = (int (*)(unsigned long long*))dlsym(lib, "lwm_conv_prof_set");
    if (!conv || !prof_set) { fprintf(stderr, "not a profiling variant\n"); return 2; }
    const Shape shapes[] = {
        {"c128_256_res", 256, 256, 128, 128, 0, 1}, {"c128_up256", 128, 128, 128, 128, 1, 0}, {"c128_256_nores", 256, 256, 128, 128, 0, 0},
        {"c256_up128", 64, 64, 256, 256, 1, 0},     {"c256_128_res", 128, 128, 256, 256, 0, 1},
    };
    for (const Shape& s : shapes) {
        const int Ho = s.H << s.up, Wo = s.W << s.up;
        const size_t nx = (size_t)B * s.H * s.W * s.Cin, nw = (size_t)9 * s.Cin * s.Cout, ny = (size_t)B * Ho * Wo * s.Cout;
        const size_t grid = (size_t)B * Ho * Wo / 64 * (s.Cout / (s.Cin == 128 ? 128 : 256));     // (an upper bound of the tile count: one record per tile)
        float *x, *w, *bias, *res, *y;
        unsigned long long* prof;
        CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&bias, s.Cout * 4)); CK(hipMalloc(&res, ny * 4)); CK(hipMalloc(&y, ny * 4));
        CK(hipMalloc(&prof, grid * 64 * 8));
        CK(hipMemset(prof, 0, grid * 64 * 8));
        fill_f32<<<2048, 256>>>(x, nx, 1u, 1.0f); fill_f32<<<256, 256>>>(w, nw, 2u, 0.03f); fill_f32<<<1, 256>>>(bias, s.Cout, 3u, 0.1f); fill_f32<<<2048, 256>>>(res, ny, 4u, 1.0f);
        LwmConvArgs a;
        memset(&a, 0, sizeof(a));
        a.x = x; a.w = w; a.bias = bias; a.residual = s.res ? res : nullptr; a.y = y;
        a.B = B; a.Hin = s.H; a.Win = s.W; a.Cin = s.Cin; a.Cout = s.Cout; a.KH = 3; a.KW = 3; a.stride = 1; a.pad = 1; a.up_shift = s.up; a.Ho = Ho; a.Wo = Wo;
        prof_set(nullptr);
        conv(&a, nullptr);                      // warm
        CK(hipDeviceSynchronize());
        prof_set(prof);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        if (conv(&a, nullptr)) return 3;
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(grid * 64);
        CK(hipMemcpy(h.data(), prof, grid * 64 * 8, hipMemcpyDeviceToHost));
        const std::string path = std::string(argv[2]) + "/" + s.name + ".bin";
        FILE* f = fopen(path.c_str(), "wb");
        fwrite(h.data(), 8, h.size(), f);
        fclose(f);
        const double flop = 2.0 * B * Ho * Wo * 9.0 * s.Cin * s.Cout;
        printf("%s grid %zu  %.3f ms  %.3f of roof\n", s.name, grid, ms, flop / ms * 1e-9 / 157.3);
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(bias)); CK(hipFree(res)); CK(hipFree(y)); CK(hipFree(prof));
    }
    return 0;
}
