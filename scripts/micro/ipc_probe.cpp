// ipc_probe.cpp -- which cross-PROCESS hand-off mechanisms work on this box (two processes, one or two GPUs)?
//   ipc_probe [bytes=67108864] [gpu_of_child=0]
// The parent (A) and a forked child (B) each hipMalloc a mailbox and a flag block, exchange hipIpcMemHandles over
// pipes, and A pushes a pattern into B's mailbox with hipMemcpyAsync on a stream, then signals; B waits on ITS
// stream, checks the bytes, and acknowledges.  Signals tried:
//   M2  hipStreamWriteValue32 on the peer's (IPC-mapped) flag / hipStreamWaitValue32 on the local flag
//   M3  one-thread kernels: system-scope release store into the peer's flag / bounded acquire spin on the local one
// Build: hipcc -O2 --offload-arch=gfx950 -o scripts/micro/ipc_probe scripts/micro/ipc_probe.cpp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "[%s] %s: %s\n", g_who, #x, hipGetErrorString(e_));                 \
            exit(3);                                                                            \
        }                                                                                       \
    } while (0)
static const char* g_who = "?";

__global__ void set_flag(uint32_t* f, uint32_t v) {
    __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void wait_flag(const uint32_t* f, uint32_t v, uint32_t* err) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > 500000000LL) {   // 5 s at 100 MHz
            *err = 1;
            return;
        }
    }
}
__global__ void fill(uint32_t* p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = seed ^ (uint32_t)(i * 2654435761u);
}
__global__ void check(const uint32_t* p, size_t n, uint32_t seed, unsigned long long* bad) {
    unsigned long long b = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b += p[i] != (seed ^ (uint32_t)(i * 2654435761u));
    if (b) atomicAdd(bad, b);
}

struct Handles { hipIpcMemHandle_t box, flags; };

static void xfer(int fd_w, int fd_r, const Handles& mine, Handles& theirs) {
    if (write(fd_w, &mine, sizeof(mine)) != (ssize_t)sizeof(mine)) exit(4);
    if (read(fd_r, &theirs, sizeof(theirs)) != (ssize_t)sizeof(theirs)) exit(4);
}
static void sync_pipe(int fd_w, int fd_r) {
    char c = 1;
    if (write(fd_w, &c, 1) != 1 || read(fd_r, &c, 1) != 1) exit(4);
}

int main(int argc, char** argv) {
    const size_t bytes = argc > 1 ? strtoull(argv[1], 0, 10) : (64u << 20);
    const int child_gpu = argc > 2 ? atoi(argv[2]) : 0;
    int ab[2], ba[2];
    if (pipe(ab) || pipe(ba)) return 4;
    const pid_t pid = fork();      // BEFORE any HIP call
    const bool A = pid != 0;
    g_who = A ? "A" : "B";
    const int fd_w = A ? ab[1] : ba[1], fd_r = A ? ba[0] : ab[0];
    CK(hipSetDevice(A ? 0 : child_gpu));
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, A ? 0 : child_gpu));
    uint32_t *box, *flags, *err;
    unsigned long long* bad;
    CK(hipMalloc(&box, bytes));
    // flags: uncached device memory when the runtime offers it (a flag written by ANOTHER GPU must not be served from this
    // GPU's L2), plain device memory otherwise
    const char* flag_kind = "uncached";
    if (hipExtMallocWithFlags((void**)&flags, 4096, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        flag_kind = "plain";
        CK(hipMalloc(&flags, 4096));
    }
    CK(hipMemset(flags, 0, 4096));
    CK(hipMalloc(&err, 4));
    CK(hipMemset(err, 0, 4));
    CK(hipMalloc(&bad, 8));
    CK(hipMemset(bad, 0, 8));
    CK(hipDeviceSynchronize());
    Handles mine, theirs;
    CK(hipIpcGetMemHandle(&mine.box, box));
    hipError_t fe = hipIpcGetMemHandle(&mine.flags, flags);
    if (fe != hipSuccess) {      // uncached memory not exportable: fall back to plain
        (void)hipGetLastError();
        CK(hipFree(flags));
        flag_kind = "plain (uncached not exportable)";
        CK(hipMalloc(&flags, 4096));
        CK(hipMemset(flags, 0, 4096));
        CK(hipDeviceSynchronize());
        CK(hipIpcGetMemHandle(&mine.flags, flags));
    }
    xfer(fd_w, fd_r, mine, theirs);
    uint32_t *pbox, *pflags;
    CK(hipIpcOpenMemHandle((void**)&pbox, theirs.box, hipIpcMemLazyEnablePeerAccess));
    CK(hipIpcOpenMemHandle((void**)&pflags, theirs.flags, hipIpcMemLazyEnablePeerAccess));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    printf("[%s] pid %d gpu %d: CanUseStreamWaitValue %d, flags %s, handles opened\n", g_who, (int)getpid(), A ? 0 : child_gpu, can, flag_kind);
    fflush(stdout);
    uint32_t* src = nullptr;
    if (A) CK(hipMalloc(&src, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int mech = 2; mech <= 3; ++mech) {
        sync_pipe(fd_w, fd_r);
        const uint32_t seq = (uint32_t)mech * 10, seed = 0x1234u * mech;
        // flag[0] on B: "data ready"; flag[1] on A: "ack"
        bool ok = true;
        if (A) {
            fill<<<1024, 256, 0, s>>>(src, bytes / 4, seed);
            CK(hipEventRecord(e0, s));
            CK(hipMemcpyAsync(pbox, src, bytes, hipMemcpyDeviceToDevice, s));
            CK(hipEventRecord(e1, s));
            if (mech == 2) {
                hipError_t e = hipStreamWriteValue32(s, pflags, seq, 0);
                if (e != hipSuccess) { printf("[A] M2 hipStreamWriteValue32 on the peer flag: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); ok = false; }
                e = ok ? hipStreamWaitValue32(s, flags + 1, seq, hipStreamWaitValueGte, 0xffffffffu) : e;
                if (ok && e != hipSuccess) { printf("[A] M2 hipStreamWaitValue32: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); ok = false; }
            } else {
                set_flag<<<1, 1, 0, s>>>(pflags, seq);
                wait_flag<<<1, 1, 0, s>>>(flags + 1, seq, err);
            }
        } else {
            if (mech == 2) {
                hipError_t e = hipStreamWaitValue32(s, flags, seq, hipStreamWaitValueGte, 0xffffffffu);
                if (e != hipSuccess) { printf("[B] M2 hipStreamWaitValue32: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); ok = false; }
            } else {
                wait_flag<<<1, 1, 0, s>>>(flags, seq, err);
            }
            check<<<1024, 256, 0, s>>>(box, bytes / 4, seed, bad);
            if (mech == 2) {
                hipError_t e = ok ? hipStreamWriteValue32(s, pflags + 1, seq, 0) : hipErrorUnknown;
                if (ok && e != hipSuccess) { printf("[B] M2 hipStreamWriteValue32: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); ok = false; }
            } else {
                set_flag<<<1, 1, 0, s>>>(pflags + 1, seq);
            }
        }
        hipError_t se = hipStreamSynchronize(s);
        uint32_t herr = 0;
        unsigned long long hbad = 0;
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost));
        CK(hipMemset(err, 0, 4));
        CK(hipMemset(bad, 0, 8));
        float ms = 0;
        if (A) (void)hipEventElapsedTime(&ms, e0, e1);
        printf("[%s] M%d %s: api %s, sync %s, spin timeout %u, wrong words %llu%s\n", g_who, mech,
               mech == 2 ? "stream write/wait value" : "flag kernels", ok ? "ok" : "FAILED", hipGetErrorString(se), herr, hbad,
               A ? "" : " (the pushed bytes, checked after the wait)");
        if (A && ms > 0) printf("[A] push of %zu bytes into the peer's mailbox: %.3f ms = %.1f GB/s\n", bytes, ms, bytes / ms / 1e6);
        fflush(stdout);
    }
    sync_pipe(fd_w, fd_r);
    CK(hipIpcCloseMemHandle(pbox));
    CK(hipIpcCloseMemHandle(pflags));
    sync_pipe(fd_w, fd_r);
    if (A) {
        int st = 0;
        waitpid(pid, &st, 0);
        printf("[A] child exit status %d\n", WEXITSTATUS(st));
    }
    return 0;
}
