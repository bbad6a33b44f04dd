// Checks the operand / result layout of v_mfma_f32_32x32x16_bf16 assumed by fused.hip:
//   A: lane l (r = l&31, h = l>>5) holds A[row r][k = 8h + j], j = 0..7
//   B: lane l holds B[k = 8h + j][col r]
//   D: register reg of lane l is D[row (reg&3) + 8*(reg>>2) + 4*h][col r]
// with exact small-integer data (asymmetric A and B).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ unsigned short f2bf(float f) { return (unsigned short)(__float_as_uint(f) >> 16); }
__global__ void k(float *out) {
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  s16x8 a, b;
  for (int j = 0; j < 8; j++) {
    int kk = 8 * h + j;
    a[j] = (short)f2bf((float)(1 + (r % 5) + 2 * (kk % 3)));   // A[r][kk]
    b[j] = (short)f2bf((float)(1 + (kk % 4) + 3 * (r % 7)));   // B[kk][r]
  }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  for (int g = 0; g < 16; g++) out[l * 16 + g] = acc[g];
}
int main() {
  float *d, hbuf[64 * 16];
  if (hipMalloc(&d, sizeof(hbuf)) != hipSuccess) { printf("no device\n"); return 1; }
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(hbuf, d, sizeof(hbuf), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++)
    for (int g = 0; g < 16; g++) {
      int r = l & 31, h = l >> 5, row = (g & 3) + 8 * (g >> 2) + 4 * h, col = r;
      float want = 0;
      for (int kk = 0; kk < 16; kk++) want += (float)(1 + (row % 5) + 2 * (kk % 3)) * (float)(1 + (kk % 4) + 3 * (col % 7));
      if (hbuf[l * 16 + g] != want) { if (bad < 6) printf("lane %d reg %d got %g want %g\n", l, g, hbuf[l * 16 + g], want); bad++; }
    }
  printf("mfma_32x32x16_bf16 layout check: %s (%d mismatches)\n", bad ? "MISMATCH" : "as assumed", bad);
  return bad != 0;
}
