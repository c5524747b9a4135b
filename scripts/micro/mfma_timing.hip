// Micro-benchmark (not product): cycles per v_mfma_f32_32x32x16_bf16 for a single wave per SIMD (256
// threads) or two (512), as a function of the number of independent accumulators and of what is
// issued between the MFMAs (nothing / one ds_read_b128 / two ds_read_b64_tr_b16 / k plain VALU).
#include "wave_ops.h"
#include "attn_common.h"
using namespace lwm;

template <int NACC, int MODE, int NVALU>
__global__ __launch_bounds__(512) void mfma_time(unsigned long long* out, float* sink, int iters) {
    lds_t lds = dyn_lds();
    const int tid = thread_idx(), lane = tid & 63;
    for (int c = tid; c < 64 * 16; c += (int)blockDim.x) {
        u32x4 v = {(uint32_t)c, 1u, 2u, 3u};
        lds_write_b128(lds + tile_off(c >> 4, c & 15), v);
    }
    block_sync();
    RowFragAddr ra = frag_rows_addr(lds, 0, lane & 31, lane >> 5);
    TrFragAddr ta = frag_tr_addr(lds, lane);
    bf16x8 b = zero_bf16x8(), a0 = zero_bf16x8();
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = zero_f32x16();
    float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    constexpr int R = 4;
    bf16x8 ring[R];
    const uint32_t base = opaque(ra.a[0]), lo0 = opaque(ta.lo[0]), up0 = opaque(ta.up[0]);
    auto load = [&](int g) {
        if (MODE == 1) ring[g % R] = lds_read_b128(row_frag_at(base, g & 7) + (g >> 3) * 32 * kRowBytes);
        if (MODE == 2) ring[g % R] = read_tr_frag_x(lo0, up0, g & 3, 16 * (g >> 2) * kRowBytes);
    };
    block_sync();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE) for (int g = 0; g < R - 1; ++g) load(g);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (MODE && g + R - 1 < 16) load(g + R - 1);
            sched_fence();
            acc[g % NACC] = mfma_32x32x16(MODE ? ring[g % R] : a0, b, acc[g % NACC]);
#pragma unroll
            for (int k = 0; k < NVALU; ++k) v[k & 7] = fmaf(v[k & 7], 1.0001f, 0.5f);
            sched_fence();
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    for (int k = 0; k < 8; ++k) s += v[k];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[tid >> 6] = t1 - t0;
    if (s == 12345.f) sink[0] = s;
}

template <int NACC, int MODE, int NVALU>
static void run1(unsigned long long* out, float* sink, int threads, int iters, hipStream_t st) {
    hipLaunchKernelGGL((mfma_time<NACC, MODE, NVALU>), dim3(1), dim3(threads), 64 * 256, st, out, sink, iters);
}

extern "C" int mfma_time_run(int nacc, int mode, int nvalu, int threads, int iters, void* out, void* sink, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* o = (unsigned long long*)out;
    float* s = (float*)sink;
#define CASE(A, M, V) if (nacc == A && mode == M && nvalu == V) { run1<A, M, V>(o, s, threads, iters, st); return (int)hipGetLastError(); }
    CASE(1, 0, 0) CASE(2, 0, 0) CASE(4, 0, 0)
    CASE(1, 1, 0) CASE(2, 1, 0) CASE(4, 1, 0)
    CASE(2, 2, 0) CASE(4, 2, 0)
    CASE(4, 0, 2) CASE(4, 0, 4) CASE(4, 0, 6) CASE(4, 0, 8)
    CASE(4, 1, 2) CASE(4, 1, 4) CASE(4, 1, 6)
    CASE(2, 1, 4)
    return -1;
}
