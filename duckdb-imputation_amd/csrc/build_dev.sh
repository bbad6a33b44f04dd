#!/bin/sh
# Dev-only: libcofactor_hip_dev.so with -DCOFACTOR_DEV_ABLATE (tests/tools/cat_ablate.py).
set -e
cd "$(dirname "$0")"
mkdir -p build_dev
for f in gram cat fused fused2 ring sparse predict; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCOFACTOR_DEV_ABLATE -I../../include -c $f.hip -o build_dev/$f.o & done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCOFACTOR_DEV_ABLATE -I../../include -x hip -c api.cpp -o build_dev/api.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCOFACTOR_DEV_ABLATE -I../../include -x hip -c ring_api.cpp -o build_dev/ring_api.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -x c++ -c triple.cpp -o build_dev/triple.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -x c++ -c ml.cpp -o build_dev/ml.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../cofactor_hip/libcofactor_hip_dev.so build_dev/gram.o build_dev/cat.o build_dev/fused.o build_dev/fused2.o build_dev/ring.o build_dev/sparse.o build_dev/ring_api.o build_dev/api.o build_dev/triple.o build_dev/ml.o build_dev/predict.o
