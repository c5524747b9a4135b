#!/bin/bash
# Build a variant of the VQGAN TU for A/B runs:  scripts/ab_build_vqgan.sh <name> <extra hipcc flags...>
# -> build/ab/liblwmv_<name>.so (attention TU = the product's build/lwm_hip.o)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mcode-object-version=5 -fPIC -c -ffp-contract=off "$@" \
    -I include -I lwm_amd/csrc lwm_amd/csrc/lwm_vqgan.hip -o build/ab/lwm_vqgan_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/lwm_hip.o build/ab/lwm_vqgan_$name.o -o build/ab/liblwmv_$name.so
echo build/ab/liblwmv_$name.so
