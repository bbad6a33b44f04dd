// Microbenchmark: LDS atomic-add throughput on the device it runs on, by type and address pattern.
// Prints lane-adds per clock per CU (assuming 2.4 GHz nominal; also prints ns).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <typename T> __device__ T one();
template <> __device__ unsigned one<unsigned>() { return 1u; }
template <> __device__ float one<float>() { return 1.0f; }
template <> __device__ double one<double>() { return 1.0; }
template <typename T> __device__ void add(T* p, T v) { unsafeAtomicAdd(p, v); }
template <> __device__ void add<unsigned>(unsigned* p, unsigned v) { atomicAdd(p, v); }

template <typename T, int ITER>
__global__ __launch_bounds__(1024) void k(const int* __restrict__ idx, int nidx, T* out) {
  __shared__ T tab[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = T(0);
  __syncthreads();
  int base = (blockIdx.x * blockDim.x + threadIdx.x) % nidx;
  int my[8];
  for (int j = 0; j < 8; j++) my[j] = idx[(base + j * 1031) % nidx];
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int j = 0; j < 8; j++) add<T>(&tab[(my[j] + it) & 4095], one<T>());
  }
  __syncthreads();
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = tab[threadIdx.x];
}

template <typename T>
void run(const char* name, const std::vector<int>& h, const char* pat) {
  int* d; T* o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 512 * 64 * sizeof(T));
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  constexpr int ITER = 256;
  hipLaunchKernelGGL((k<T, ITER>), dim3(512), dim3(1024), 0, 0, d, (int)h.size(), o);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<T, ITER>), dim3(512), dim3(1024), 0, 0, d, (int)h.size(), o);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double adds = 512.0 * 1024 * ITER * 8;
  printf("%-4s %-28s %8.3f ms  %7.2f lane-adds/clk/CU (256 CUs @2.4GHz)\n", name, pat, ms,
         adds / (ms * 1e-3) / 256 / 2.4e9);
  hipFree(d); hipFree(o);
}

int main() {
  std::vector<int> uniq(65536), k16(65536), k16s10(65536), rnd256(65536), same(65536, 0);
  unsigned s = 12345;
  for (int i = 0; i < 65536; i++) {
    s = s * 1664525u + 1013904223u;
    uniq[i] = i % 4096;
    int r = (s >> 8);
    k16[i] = r % 16;
    k16s10[i] = (r % 16) * 10;
    rnd256[i] = r % 256;
  }
  // uniq is read with base = thread id so lanes of a wave hit 64 consecutive cells
  struct { const char* n; std::vector<int>* v; } pats[] = {
      {"consecutive cells", &uniq}, {"16 random cells", &k16}, {"16 cells stride 10", &k16s10},
      {"256 random cells", &rnd256}, {"one cell", &same}};
  for (auto& p : pats) {
    run<unsigned>("u32", *p.v, p.n);
    run<float>("f32", *p.v, p.n);
    run<double>("f64", *p.v, p.n);
  }
  return 0;
}
