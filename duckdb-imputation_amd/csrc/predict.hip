// Per-row predictors over device-resident columns for the models ml.cpp trains (SURVEY.md §8f
// N2): out[row] = W . [1, x, onehot(keys)] (+ Gaussian noise) for linreg_predict
// (ML::linreg_impute, ML/regression.cpp:397-508), argmax over classes for lda_predict
// (LDA_impute, ML/lda.cpp:421-590).  HBM-bound: every input column is read once, coalesced, one
// row per lane; the model (a few KB) sits in LDS.
#include "device.hpp"

namespace cofactor {

namespace {

constexpr int PREDICT_THREADS = 256;

// counter-based generator: two uniforms per (seed, row), independent of the launch geometry
__device__ inline unsigned long long mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <bool ARGMAX>
__global__ __launch_bounds__(PREDICT_THREADS) void predict_kernel(
    NumCols num, CatCols cat, int F, int M, int C, int KT, const int32_t *__restrict__ kbegin,
    const int32_t *__restrict__ keys, const double *__restrict__ W, int w_in_lds,
    const uint8_t *__restrict__ mask, uint64_t rows, float *__restrict__ out_f,
    int32_t *__restrict__ out_i, const int32_t *__restrict__ labels, int noise, double noise_sd,
    unsigned long long seed) {
  extern __shared__ double smem[];
  // LDS: [W (C * P doubles) if it fits][keys KT][kbegin M+1][x tile F x 256][slot tile M x 256]
  const int P = 1 + F + KT;
  double *w_l = smem;
  int32_t *keys_l = reinterpret_cast<int32_t *>(smem + (w_in_lds ? (size_t)C * P : 0));
  int32_t *kb_l = keys_l + KT;
  float *x_l = reinterpret_cast<float *>(kb_l + M + 1);
  int32_t *s_l = reinterpret_cast<int32_t *>(x_l + (size_t)F * PREDICT_THREADS);
  const int tid = threadIdx.x;
  if (w_in_lds)
    for (int i = tid; i < C * P; i += PREDICT_THREADS) w_l[i] = W[i];
  for (int i = tid; i < KT; i += PREDICT_THREADS) keys_l[i] = keys[i];
  for (int i = tid; i <= M; i += PREDICT_THREADS) kb_l[i] = kbegin[i];
  __syncthreads();
  const double *w = w_in_lds ? w_l : W;

  for (uint64_t row = (uint64_t)blockIdx.x * PREDICT_THREADS + tid; row < rows;
       row += (uint64_t)gridDim.x * PREDICT_THREADS) {
    if (mask && !mask[row]) continue;
    for (int f = 0; f < F; f++) x_l[f * PREDICT_THREADS + tid] = __builtin_nontemporal_load(num.p[f] + row);
    for (int c = 0; c < M; c++) {
      const int32_t key = __builtin_nontemporal_load(cat.p[c] + row);
      int lo = kb_l[c], hi = kb_l[c + 1];
      const int end = hi;
      while (lo < hi) {  // keys ascending within a column
        const int mid = (lo + hi) >> 1;
        if (keys_l[mid] < key) lo = mid + 1; else hi = mid;
      }
      s_l[c * PREDICT_THREADS + tid] = (lo < end && keys_l[lo] == key) ? lo : -1;
    }
    double best = 0;
    int arg = 0;
    for (int k = 0; k < C; k++) {
      const double *wk = w + (size_t)k * P;
      double v = wk[0];
      for (int f = 0; f < F; f++) v += wk[1 + f] * (double)x_l[f * PREDICT_THREADS + tid];
      for (int c = 0; c < M; c++) {
        const int s = s_l[c * PREDICT_THREADS + tid];
        if (s >= 0) v += wk[1 + F + s];
      }
      if (k == 0 || v > best) { best = v; arg = k; }
    }
    if (ARGMAX) {
      out_i[row] = labels ? labels[arg] : arg;
    } else {
      if (noise) {
        const unsigned long long h = mix64(seed + 0x9E3779B97F4A7C15ull * (row + 1));
        const double u1 = ((double)(h >> 32) + 1.0) * (1.0 / 4294967296.0);  // (0, 1]
        const double u2 = (double)(h & 0xFFFFFFFFull) * (1.0 / 4294967296.0);
        best += noise_sd * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
      }
      out_f[row] = (float)best;
    }
  }
}

}  // namespace

size_t predict_lds_bytes(int F, int M, int C, int KT, size_t lds_limit, int *w_in_lds) {
  const size_t fixed = ((size_t)KT + M + 1) * 4 + ((size_t)F + M) * PREDICT_THREADS * 4;
  const size_t wbytes = (size_t)C * (1 + F + KT) * 8;
  *w_in_lds = fixed + wbytes <= lds_limit;
  return fixed + (*w_in_lds ? wbytes : 0) + 8;
}

hipError_t launch_predict(bool argmax, const NumCols &num, const CatCols &cat, int F, int M, int C,
                          int KT, const int32_t *kbegin, const int32_t *keys, const double *W,
                          const uint8_t *mask, uint64_t rows, float *out_f, int32_t *out_i,
                          const int32_t *labels, int noise, double noise_sd,
                          unsigned long long seed, int grid, size_t lds_limit, hipStream_t stream) {
  if (rows == 0) return hipSuccess;
  int w_in_lds = 0;
  const size_t lds = predict_lds_bytes(F, M, C, KT, lds_limit, &w_in_lds);
  if (lds > lds_limit) return hipErrorInvalidValue;
  const uint64_t need = (rows + PREDICT_THREADS - 1) / PREDICT_THREADS;
  if ((uint64_t)grid > need) grid = (int)need;
  hipError_t e;
  if (argmax) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&predict_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    predict_kernel<true><<<grid, PREDICT_THREADS, lds, stream>>>(
        num, cat, F, M, C, KT, kbegin, keys, W, w_in_lds, mask, rows, out_f, out_i, labels, noise,
        noise_sd, seed);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&predict_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    predict_kernel<false><<<grid, PREDICT_THREADS, lds, stream>>>(
        num, cat, F, M, C, KT, kbegin, keys, W, w_in_lds, mask, rows, out_f, out_i, labels, noise,
        noise_sd, seed);
  }
  return hipGetLastError();
}

}  // namespace cofactor
