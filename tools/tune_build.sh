#!/bin/bash
# Builds tuning variants of the library (CPU side): tools/tune_build.sh <name> <-D flags...>  ->  volumerenderer_amd/tune/libvr_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../volumerenderer_amd/csrc"
mkdir -p ../tune
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-result -shared "$@" -o ../tune/libvr_$name.so vr_kernels.hip vr_build.hip vr_refine.hip vr_capi.hip vr_host.cpp vr_image_io.cpp
