#!/bin/sh
# Builds duckdb-imputation_amd/cofactor_hip/libcofactor_hip_<name>.so with fused.hip taken from git
# revision <rev> and every other object from the current build: the "A" of an A/B run
# (tests/tools/ab_fused.sh).   sh tests/tools/build_variant.sh <rev> <name>
set -e
cd "$(dirname "$0")/../../duckdb-imputation_amd/csrc"
mkdir -p build_variant
git show "$1:duckdb-imputation_amd/csrc/fused.hip" > build_variant/fused_$2.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -c build_variant/fused_$2.hip -o build_variant/fused_$2.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../cofactor_hip/libcofactor_hip_$2.so \
  build/gram.o build/cat.o build_variant/fused_$2.o build/predict.o build/api.o build/triple.o build/ml.o
echo built libcofactor_hip_$2.so
