#!/bin/bash
# Instrumented build of the attention TU (per-phase s_memtime stamps, see ProfAcc in
# lwm_amd/csrc/attn_common.h).  Never loaded by the package; only scripts/prof_phases.py uses it.
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mcode-object-version=5 -shared -fPIC \
    -fno-slp-vectorize -DLWM_PROF -I include -I lwm_amd/csrc lwm_amd/csrc/lwm_hip.hip -o scripts/liblwm_prof.so
echo "built scripts/liblwm_prof.so"
