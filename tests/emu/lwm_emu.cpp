// lwm_emu.cpp -- host-emulated build of the SAME kernel headers and C ABI as
// liblwm_hip.so (TEST INFRASTRUCTURE ONLY; see tests/emu/wave_ops.h).
// Build: clang++ -O2 -std=c++17 -shared -fPIC -I tests/emu -I lwm_amd/csrc -I include
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "emu/wave_ops.h"
#include "emu/launch.h"
#include "lwm_hip.h"
#include "attn_common.h"
#include "attn_fwd.h"
#include "attn_fwd64.h"
#include "attn_bwd.h"
#include "attn_bwd64.h"
#include "attn_decode.h"
#include "misc_kernels.h"
#include "llama_elem.h"
#include "gemv.h"
#include "gemm_wgrad.h"
#include "attn_f32.h"
#include "elem_f32.h"
#include "api.inc"
#include "api_f32.inc"
#include "vqgan_conv.h"
#include "vqgan_misc.h"
#include "vqgan_api.inc"

// The ring driver is HIP-runtime code (streams, events, RCCL): not emulated.  The emulated library still
// exports the symbols so that one ctypes binding serves both builds.
extern "C" {
#define LWM_EMU_NO_RING(ret, name, ...) ret name(__VA_ARGS__) { return (ret)lwm::fail(LWM_EUNSUPPORTED, "%s", #name ": not in the host emulation"); }
LWM_EMU_NO_RING(int, lwm_ring_create, void*, int32_t, int32_t, void*, LwmRing**)
LWM_EMU_NO_RING(int, lwm_ring_unique_id, void*)
LWM_EMU_NO_RING(int, lwm_ring_create_from_id, const void*, int32_t, int32_t, void*, LwmRing**)
LWM_EMU_NO_RING(int, lwm_ring_create_transport, const LwmRingTransport*, int32_t, int32_t, void*, LwmRing**)
LWM_EMU_NO_RING(int, lwm_ring_destroy, LwmRing*)
LWM_EMU_NO_RING(int64_t, lwm_ring_workspace_bytes, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t)
LWM_EMU_NO_RING(int, lwm_ring_attn_fwd, LwmRing*, const LwmRingArgs*, void*)
LWM_EMU_NO_RING(int, lwm_ring_attn_bwd, LwmRing*, const LwmRingArgs*, void*)
LWM_EMU_NO_RING(int64_t, lwm_ring_bytes_sent, const LwmRing*)
LWM_EMU_NO_RING(int, lwm_ring_last_form, const LwmRing*)
LWM_EMU_NO_RING(int64_t, lwm_ring_kv_keep_bytes, int32_t, int32_t, int32_t, int32_t, int32_t)
LWM_EMU_NO_RING(int64_t, lwm_ring_planned_bytes_table, const int32_t*, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t)
LWM_EMU_NO_RING(int64_t, lwm_ring_planned_bytes, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t)
LWM_EMU_NO_RING(int, lwm_ring_selftest, LwmRing*, const void*, void*, int64_t, void*)
LWM_EMU_NO_RING(int, lwm_ring_set_fetch_groups, LwmRing*, int32_t)
LWM_EMU_NO_RING(int, lwm_ring_fetch_timeline, LwmRing*, float*, float*, int32_t)
LWM_EMU_NO_RING(int64_t, lwm_ring_ipc_info_bytes, void)
LWM_EMU_NO_RING(int, lwm_ring_ipc_export, int32_t, int32_t, int64_t, int32_t, void*, LwmRingIpc**)
LWM_EMU_NO_RING(int, lwm_ring_ipc_connect, LwmRingIpc*, const void*)
LWM_EMU_NO_RING(int, lwm_ring_create_ipc, LwmRingIpc*, void*, LwmRing**)
LWM_EMU_NO_RING(int, lwm_ring_ipc_destroy, LwmRingIpc*)
}
