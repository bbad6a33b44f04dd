#!/bin/sh
# Dev-only: libcofactor_hip_f3dev.so = the current build with fused3.hip recompiled with the phase
# switches (-DCOFACTOR_DEV_ABLATE) and only the 10_10 instantiation.   sh tests/tools/build_f3dev.sh [extra flags]
set -e
cd "$(dirname "$0")/../../duckdb-imputation_amd/csrc"
mkdir -p build_variant
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -DCOFACTOR_DEV_ABLATE -DF3_DEV_ONLY_10_10 "$@" -c fused3.hip -o build_variant/fused3_dev.o
OBJS=""
for o in gram cat fused fused2 ring sparse predict api ring_api triple ml; do OBJS="$OBJS build/$o.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../cofactor_hip/libcofactor_hip_f3dev.so $OBJS build_variant/fused3_dev.o
echo built libcofactor_hip_f3dev.so
