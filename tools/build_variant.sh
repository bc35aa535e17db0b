#!/bin/bash
# Experiment builds of the HIP library: tools/build_variant.sh <name> [extra hipcc flags].  Output: variants/librt_hip_<name>.so (not tracked);
# run with RT_HIP_LIB=../../variants/librt_hip_<name>.so python bench.py --no-extras --no-cpu-baseline.  -DRT_MINIMAL builds only the
# plain walk kernels (half a minute instead of three).
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
N=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared "$@" -o variants/librt_hip_$N.so ray_tracing_amd/csrc/rt_hip.hip
ls -la variants/librt_hip_$N.so
