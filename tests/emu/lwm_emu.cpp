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
#include "attn_bwd.h"
#include "attn_bwd_fused.h"
#include "attn_decode.h"
#include "misc_kernels.h"
#include "llama_elem.h"
#include "api.inc"
#include "vqgan_conv.h"
#include "vqgan_misc.h"
#include "vqgan_api.inc"
