#!/bin/bash
# Build a variant of the attention TU for A/B runs:  scripts/ab_build.sh <name> <extra hipcc flags...>
# -> build/ab/liblwm_<name>.so ; select on the GPU box with LWM_HIP_LIB=build/ab/liblwm_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mcode-object-version=5 -fPIC -c -fno-slp-vectorize "$@" \
    -I include -I lwm_amd/csrc lwm_amd/csrc/lwm_hip.hip -o build/ab/lwm_hip_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/ab/lwm_hip_$name.o build/lwm_vqgan.o -o build/ab/liblwm_$name.so
echo build/ab/liblwm_$name.so
