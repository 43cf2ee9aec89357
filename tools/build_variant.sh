#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags...]  ->  gpurun_variants/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -Wno-parentheses-equality "$@" digiham_amd/csrc/engine.hip -o variants/lib_$NAME.so
echo built variants/lib_$NAME.so
