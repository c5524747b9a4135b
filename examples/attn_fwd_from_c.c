/* Calling the hot path straight from C: no Python, no torch -- only the HIP runtime for memory and a
 * stream, and liblwm_hip.so through include/lwm_hip.h.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/attn_fwd_from_c.c \
 *       lwm_amd/liblwm_hip.so -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/lwm_amd -o attn_fwd_from_c
 *
 * One causal attention block, B=1, S=1024, H=4, D=128, bf16: out and lse come back in the layouts of
 * include/lwm_hip.h ([B,S,H,D] bf16 and [B,H,S] f32).  Exit status 0 on success.                      */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lwm_hip.h"

static uint16_t to_bf16(float f) {           /* round to nearest even */
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(void) {
    const int B = 1, S = 1024, H = 4, D = 128;
    const size_t n = (size_t)B * S * H * D;
    uint16_t* host = (uint16_t*)malloc(n * 2);
    void *q, *k, *v, *out;
    float* lse;
    hipStream_t stream;
    LwmAttnArgs a;
    LwmTensor4 t;
    int rc;
    size_t i;

    if (lwm_sizeof(0) != (int)sizeof(LwmAttnArgs)) return 3;      /* header and library agree */
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_HIP(hipMalloc(&q, n * 2));
    CHECK_HIP(hipMalloc(&k, n * 2));
    CHECK_HIP(hipMalloc(&v, n * 2));
    CHECK_HIP(hipMalloc(&out, n * 2));
    CHECK_HIP(hipMalloc((void**)&lse, (size_t)B * H * S * sizeof(float)));
    srand(1);
    for (i = 0; i < n; ++i) host[i] = to_bf16((float)rand() / RAND_MAX - 0.5f);
    CHECK_HIP(hipMemcpy(q, host, n * 2, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(k, host, n * 2, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(v, host, n * 2, hipMemcpyHostToDevice));

    memset(&a, 0, sizeof a);
    t.stride_b = (int64_t)S * H * D; t.stride_s = (int64_t)H * D; t.stride_h = D;     /* strides in elements */
    t.ptr = q;   a.q = t;
    t.ptr = k;   a.k = t;
    t.ptr = v;   a.v = t;
    t.ptr = out; a.out = t;
    a.lse = lse;
    a.B = B; a.H = H; a.Sq = S; a.Sk = S; a.D = D;
    a.q_start = 0; a.k_start = 0;                 /* global positions of row 0 of the q / k block */
    a.scale = 1.0f / sqrtf((float)D);
    a.causal = 1; a.carry_in = 0; a.final_out = 1;

    rc = lwm_attn_fwd(&a, stream);                /* enqueues on `stream`; never synchronises */
    if (rc != LWM_OK) { fprintf(stderr, "lwm_attn_fwd: %s\n", lwm_last_error()); return 4; }
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(host, out, 16, hipMemcpyDeviceToHost));
    printf("ok: out[0,0,0,0..1] = %04x %04x (row 0 sees only key 0: equals v[0,0,0,:])\n", host[0], host[1]);
    return 0;
}
