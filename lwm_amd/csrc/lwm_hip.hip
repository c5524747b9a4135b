// lwm_hip.hip -- translation unit of liblwm_hip.so (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC (see build.py).
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <unistd.h>
#include "wave_ops.h"
#include "launch.h"
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
#include "ring_driver.inc"
#include "ring_ipc.inc"
