// f4_stream.hip -- the two phases of attn_fwd64.h timed in isolation (one wave per SIMD, 256 workgroups of 4 waves):
// which part of a phase costs what.  Each variant runs REPS half steps back to back on LDS-resident tiles (no DMA, no
// barrier, no masks) and reports s_memtime cycles per phase call (16 MFMAs = 512 cycles of matrix pipe).
// Build: hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -I include -I lwm_amd/csrc -o scripts/micro/f4_stream scripts/micro/f4_stream.hip
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "wave_ops.h"
#include "launch.h"
#include "lwm_hip.h"
#include "attn_common.h"
#include "attn_fwd.h"
#include "attn_fwd64.h"
using namespace lwm;

constexpr int REPS = 200;

template <int MODE>
__global__ __launch_bounds__(256) void stream_kernel(const bf16_t* src, unsigned long long* out, float* sink) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx(), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = wave_uniform(tid >> 6);
    for (int i = tid; i < 4 * kF4TileBytes / 16; i += 256)
        lds_write_b128(lds + 16 * i, global_load_b128(src + 8 * (i & 4095)));
    block_sync();
    F4Ctx cx;
    cx.lds = lds; cx.tid = tid; cx.lane = lane; cx.hi = hi; cx.wave = wave;
    for (int s = 0; s < 8; ++s) cx.ka[s] = lds + tile_off(l31, 2 * s + hi);
    {
        const TrFragAddr t = frag_tr_addr(lds + 2 * kF4TileBytes, lane);
        for (int db = 0; db < 4; ++db) { cx.vlo[db] = t.lo[db]; cx.vup[db] = t.up[db]; }
    }
    cx.c = 0.127f; cx.thr_on = 60.f;
    for (int q = 0; q < 2; ++q) { cx.thr[q] = 1e30f; cx.mref[q] = 0.f; cx.nbase[q] = -1.0f; cx.lsum[q] = 0.f; }
    bf16x8 qf[2][8];
    for (int q = 0; q < 2; ++q)
        for (int s = 0; s < 8; ++s) qf[q][s] = f4_load_agpr(src + (size_t)(tid * 16 + q * 8 + s) * 8);
    f4_load_agpr_wait(qf);
    f32x16 acc[2][4], sA[2], sB[2];
    float tt[2][16], ps[2][8], mx[2] = {0.f, 0.f};
    bf16x8 pb[2][2], kfr[4], vfr[4];
    for (int q = 0; q < 2; ++q) {
        for (int d = 0; d < 4; ++d) acc[q][d] = zero_f32x16();
        sA[q] = zero_f32x16(); sB[q] = zero_f32x16();
        for (int r = 0; r < 16; ++r) tt[q][r] = -3.0f - 0.01f * r;
        for (int r = 0; r < 8; ++r) ps[q][r] = 0.f;
        for (int t = 0; t < 2; ++t) pb[q][t] = zero_bf16x8();
    }
    for (int j = 0; j < 4; ++j) { kfr[j] = f4_kread<0>(cx, j & 3); vfr[j] = f4_vread<0>(cx, j & 3); }
    F4Stage st = {};
    F4Dma dm = {};
    block_sync();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < REPS; ++it) {
        if (MODE == 0) f4_phase1<0, true, false, -1, -1>(cx, qf, sB, tt, ps, pb, kfr, vfr, st, dm);      // S MFMAs + K reads
        if (MODE == 1) f4_phase1<0, false, true, -1, -1>(cx, qf, sB, tt, ps, pb, kfr, vfr, st, dm);     // finish fillers only
        if (MODE == 2) f4_phase1<0, true, true, -1, -1>(cx, qf, sB, tt, ps, pb, kfr, vfr, st, dm);      // both
        if (MODE == 3) f4_phase1<0, true, true, 0, -1>(cx, qf, sB, tt, ps, pb, kfr, vfr, st, dm);       // both + V prefetch
        if (MODE == 4) f4_phase2<0, true, false, -1, -1>(cx, pb, acc, ps, sB, tt, mx, kfr, vfr, st, dm); // PV MFMAs + V reads + adds
        if (MODE == 5) f4_phase2<0, false, true, -1, -1>(cx, pb, acc, ps, sB, tt, mx, kfr, vfr, st, dm); // max + fma only
        if (MODE == 6) f4_phase2<0, true, true, -1, -1>(cx, pb, acc, ps, sB, tt, mx, kfr, vfr, st, dm);  // both
        if (MODE == 7) f4_phase2<0, true, true, 0, -1>(cx, pb, acc, ps, sB, tt, mx, kfr, vfr, st, dm);   // both + K prefetch
        if (MODE == 8) {    // a whole half step as the kernel runs it
            f4_phase1<0, true, true, 0, -1>(cx, qf, sB, tt, ps, pb, kfr, vfr, st, dm);
            f4_phase2<0, true, true, 0, -1>(cx, pb, acc, ps, sB, tt, mx, kfr, vfr, st, dm);
        }
        if (MODE <= 3 || MODE == 8)
            for (int r = 0; r < 16; ++r) { tt[0][r] = -3.0f - 0.01f * r; tt[1][r] = -2.0f - 0.01f * r; }   // keep exponents sane
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f4_mfma_settle();
    float s = mx[0] + mx[1] + cx.lsum[0] + cx.lsum[1];
    for (int q = 0; q < 2; ++q) {
        for (int d = 0; d < 4; ++d) s += acc[q][d][3];
        s += sB[q][5] + tt[q][7] + ps[q][3];
    }
    if (s == 1234.5f) *sink = s;
    if (lane == 0 && block_idx_x() == 0) out[wave] = (t1 - t0) / REPS;
}

template <int MODE>
static void run(const char* what, const bf16_t* src, unsigned long long* out, float* sink) {
    hipFuncSetAttribute((const void*)stream_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, kF4LdsBytes);
    hipLaunchKernelGGL(stream_kernel<MODE>, dim3(256), dim3(256), kF4LdsBytes, 0, src, out, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(stream_kernel<MODE>, dim3(256), dim3(256), kF4LdsBytes, 0, src, out, sink);
    hipDeviceSynchronize();
    unsigned long long h[4];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-46s cycles per call, waves 0-3: %5llu %5llu %5llu %5llu\n", what, h[0], h[1], h[2], h[3]);
}

int main() {
    bf16_t* src;
    unsigned long long* out;
    float* sink;
    hipMalloc(&src, 1 << 20);
    hipMemset(src, 0x3c, 1 << 20);
    hipMalloc(&out, 64);
    hipMalloc(&sink, 4);
    run<0>("phase 1: S MFMAs + K reads", src, out, sink);
    run<1>("phase 1: exp / cvt / pair-sum fillers only", src, out, sink);
    run<2>("phase 1: both", src, out, sink);
    run<3>("phase 1: both + V prefetch", src, out, sink);
    run<4>("phase 2: P.V MFMAs + V reads + row sums", src, out, sink);
    run<5>("phase 2: max3 + fma fillers only", src, out, sink);
    run<6>("phase 2: both", src, out, sink);
    run<7>("phase 2: both + K prefetch", src, out, sink);
    run<8>("half step: phase 1 + phase 2", src, out, sink);
    return 0;
}
