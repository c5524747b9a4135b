// probe.hip -- GPU-side confirmation of the lane maps the kernels (and the host
// emulator) assume: MFMA 32x32x16 A/B/C-D fragments, ds_read_b128 row fragments
// and ds_read_b64_tr_b16 transposed fragments through the swizzled tile image.
// TEST INFRASTRUCTURE (built by __graft_entry__.build(), used by tests/test_gpu_probe.py).
#include "wave_ops.h"
#include "attn_common.h"

using namespace lwm;

// A[32][16], B[16][32] bf16 row-major; C[32][32] f32.
__global__ __launch_bounds__(64) void probe_mfma(const bf16_t* A, const bf16_t* Bm, float* Cm) {
    int l = thread_idx(), l31 = l & 31, hi = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = A[l31 * 16 + 8 * hi + j];
        b[j] = Bm[(8 * hi + j) * 32 + l31];
    }
    f32x16 c = mfma_32x32x16(a, b, zero_f32x16());
    for (int r = 0; r < 16; ++r) Cm[cd_row(r, hi) * 32 + l31] = c[r];
}

// T[64][128] bf16 row-major -> swizzled LDS tile -> fragments.
// rows_out[lane][8]  = row fragment  (rows row0.., k-step `step`)
// cols_out[lane][8]  = transposed fragment (rows row0t.., d block d0/32)
__global__ __launch_bounds__(64) void probe_frags(const bf16_t* T, bf16_t* rows_out, bf16_t* cols_out,
                                                  int row0, int step, int row0t, int d0) {
    lds_t lds = dyn_lds();
    int l = thread_idx();
    for (int c = l; c < 64 * 16; c += 64) {
        int row = c >> 4, slot = c & 15;
        lds_write_b128(lds + tile_off(row, slot), global_load_b128(T + row * 128 + slot * 8));
    }
    block_sync();
    RowFragAddr ra = frag_rows_addr(lds, 0, l & 31, l >> 5);
    TrFragAddr ta = frag_tr_addr(lds, l);
    bf16x8 fr, fc;
    // runtime-indexed on purpose here (probe only): select by loops over constants
    for (int s = 0; s < 8; ++s)
        if (s == step) fr = lds_read_b128(ra.a[s] + row0 * kRowBytes);
    for (int db = 0; db < 4; ++db)
        if (db == d0 / 32) fc = read_tr_frag(ta, db, row0t * kRowBytes);
    for (int j = 0; j < 8; ++j) {
        rows_out[l * 8 + j] = fr[j];
        cols_out[l * 8 + j] = fc[j];
    }
}

extern "C" int probe_run_mfma(const void* A, const void* B, float* C, void* stream) {
    hipLaunchKernelGGL(probe_mfma, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)A,
                       (const bf16_t*)B, C);
    return (int)hipGetLastError();
}
extern "C" int probe_run_frags(const void* T, void* rows_out, void* cols_out, int row0, int step,
                               int row0t, int d0, void* stream) {
    hipLaunchKernelGGL(probe_frags, dim3(1), dim3(64), 64 * 256, (hipStream_t)stream,
                       (const bf16_t*)T, (bf16_t*)rows_out, (bf16_t*)cols_out, row0, step, row0t, d0);
    return (int)hipGetLastError();
}

// f32 MFMA: A[32][K] row-major, B[K][32] row-major, K even; C[32][32] =
// chain of v_mfma_f32_32x32x2_f32 over k-pairs starting from C0.  The test
// compares bitwise with the host fmaf chain in k order.
__global__ __launch_bounds__(64) void probe_mfma_f32(const float* A, const float* Bm, const float* C0,
                                                     float* Cm, int K) {
    int l = thread_idx(), l31 = l & 31, hi = l >> 5;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = C0[cd_row(r, hi) * 32 + l31];
    for (int s = 0; s < K / 2; ++s)
        c = mfma_32x32x2_f32(A[l31 * K + 2 * s + hi], Bm[(2 * s + hi) * 32 + l31], c);
    for (int r = 0; r < 16; ++r) Cm[cd_row(r, hi) * 32 + l31] = c[r];
}
extern "C" int probe_run_mfma_f32(const float* A, const float* B, const float* C0, float* C, int K,
                                  void* stream) {
    hipLaunchKernelGGL(probe_mfma_f32, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, C0, C, K);
    return (int)hipGetLastError();
}

// XCC id census: block b writes (xcc_id, a timestamp-free arrival order per XCC) -- which XCD does a
// workgroup run on, and how many workgroups of a 256-block persistent grid land on each.
__global__ __launch_bounds__(512) void probe_xcc(int* out, int* per_xcc) {
    if (thread_idx() == 0) {
        const int x = xcc_id();
        out[block_idx_x()] = x;
        atomic_add_i32(per_xcc + (x & 15), 1);
    }
}
extern "C" int probe_run_xcc(int* out, int* per_xcc, int blocks, int lds_bytes, void* stream) {
    if (lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)probe_xcc, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(probe_xcc, dim3(blocks), dim3(512), lds_bytes, (hipStream_t)stream, out, per_xcc);
    return (int)hipGetLastError();
}

