#!/bin/sh
# Builds duckdb-imputation_amd/cofactor_hip/libcofactor_hip_<name>.so with ONE kernel file
# (default fused.hip) taken from git revision <rev> and every other object from the current
# build: the "A" of a same-box A/B run.   sh tests/tools/build_variant.sh <rev> <name> [file-stem]
set -e
STEM=${3:-fused}
cd "$(dirname "$0")/../../duckdb-imputation_amd/csrc"
mkdir -p build_variant
git show "$1:duckdb-imputation_amd/csrc/$STEM.hip" > build_variant/${STEM}_$2.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -c build_variant/${STEM}_$2.hip -o build_variant/${STEM}_$2.o
OBJS=""
for o in gram cat fused predict api triple ml; do
  if [ "$o" = "$STEM" ]; then OBJS="$OBJS build_variant/${STEM}_$2.o"; else OBJS="$OBJS build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../cofactor_hip/libcofactor_hip_$2.so $OBJS
echo built libcofactor_hip_$2.so
