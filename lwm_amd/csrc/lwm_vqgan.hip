// lwm_vqgan.hip -- second translation unit of liblwm_hip.so (gfx950): the VQGAN
// primitives.  Compiled with -ffp-contract=off: their arithmetic contract
// (oracle/vqgan_ref.c) names every fused multiply-add explicitly.
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "wave_ops.h"
#include "launch.h"
#include "lwm_hip.h"
#include "attn_common.h"
#include "vqgan_conv.h"
#include "vqgan_misc.h"
#include "vqgan_api.inc"
