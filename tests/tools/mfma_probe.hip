// Prints the lane/register layout of v_mfma_f32_4x4x1_16b_f32 on the device it runs on:
// A[lane] = 100 + lane, B[lane] = 1000 + lane  ->  each D value a*b factors back into (a, b).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out) {
  int l = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(2 + l), (float)(101 + 2 * l), acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) out[l * 4 + r] = acc[r];
}
int main() {
  float *d, h[256];
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) { printf("no device\n"); return 1; }
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++)
    for (int r = 0; r < 4; r++) {
      // expected by the kernel's assumption: D[lane 4b+t][reg i] = A[lane 4b+i] * B[lane 4b+t]
      int b = l / 4, t = l % 4;
      float want = (float)(2 + 4 * b + r) * (float)(101 + 2 * (4 * b + t));
      if (h[l * 4 + r] != want) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, r, h[l * 4 + r], want); bad++; }
    }
  printf("mfma_4x4x1 layout check: %s (%d mismatches)\n", bad ? "MISMATCH" : "as assumed", bad);
  return bad != 0;
}
