// What is the fastest read-only stream on this GPU?  Sweeps loads in flight, grid size, cache
// policy and access shape over an 8 GiB buffer (the ceiling gram_kernel is compared with).
//   hipcc --offload-arch=gfx950 -O3 tests/tools/read_probe.hip -o tests/tools/read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT, bool CHUNK>
__global__ __launch_bounds__(256) void rd(const f32x4 *__restrict__ src, float *out, uint64_t n4) {
  f32x4 s[U];
  for (int u = 0; u < U; u++) s[u] = f32x4{0, 0, 0, 0};
  uint64_t i, end, stride;
  if (CHUNK) {                       // each workgroup walks its own contiguous range
    const uint64_t per = (n4 / gridDim.x) / (256 * U) * (256 * U);
    i = (uint64_t)blockIdx.x * per + threadIdx.x;
    end = (uint64_t)blockIdx.x * per + per;
    stride = 256;
  } else {
    stride = (uint64_t)gridDim.x * 256;
    i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    end = n4 / (stride * U) * (stride * U);
  }
  for (; i < end; i += U * stride)
#pragma unroll
    for (int u = 0; u < U; u++) s[u] += NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
  f32x4 t = s[0];
  for (int u = 1; u < U; u++) t += s[u];
  if (t[0] + t[1] + t[2] + t[3] == 12345.678f) out[0] = 1.f;
}

template <int U, bool NT, bool CHUNK>
void run(const f32x4 *src, float *out, uint64_t n4, int cus) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mult : {2, 4, 8, 16, 32}) {
    const int grid = cus * mult;
    hipLaunchKernelGGL((rd<U, NT, CHUNK>), dim3(grid), dim3(256), 0, 0, src, out, n4);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((rd<U, NT, CHUNK>), dim3(grid), dim3(256), 0, 0, src, out, n4);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("U=%d nt=%d chunk=%d wg/cu=%2d  %.0f GB/s\n", U, (int)NT, (int)CHUNK, mult, 5.0 * n4 * 16 / (ms * 1e-3) / 1e9);
  }
}

int main() {
  const uint64_t bytes = 8ull << 30, n4 = bytes / 16;
  f32x4 *src; float *out;
  hipMalloc((void **)&src, bytes); hipMalloc((void **)&out, 4);
  hipMemset(src, 0x3c, bytes);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  run<2, true, false>(src, out, n4, cus); run<4, true, false>(src, out, n4, cus); run<8, true, false>(src, out, n4, cus);
  run<4, false, false>(src, out, n4, cus); run<8, false, false>(src, out, n4, cus);
  run<4, true, true>(src, out, n4, cus); run<8, true, true>(src, out, n4, cus); run<8, false, true>(src, out, n4, cus);
  return 0;
}
