// Per-row predictors over device-resident columns for the models ml.cpp trains (SURVEY.md §8f
// N2): out[row] = W . [1, x, onehot(keys)] (+ Gaussian noise) for linreg_predict
// (ML::linreg_impute, ML/regression.cpp:397-508), argmax over classes for lda_predict
// (LDA_impute, ML/lda.cpp:421-590).  HBM-bound: every input column is read once, one row per
// lane; the model (a few KB) sits in LDS.  Under a row filter each wave compacts the selected
// rows first, so a sparse filter costs the bytes of the lines it touches, not idle lanes.
#include "device.hpp"

namespace cofactor {

namespace {

constexpr int PREDICT_THREADS = 256;
constexpr int PREDICT_QCAP = 64 + 256;  // per wave: < 64 rows carried over + one 256-row chunk

// offset (in doubles) of the row queues inside the dynamic LDS block
__host__ __device__ inline size_t predict_queue_offset(int F, int M, int C, int KT, int w_in_lds) {
  const size_t bytes = (w_in_lds ? (size_t)C * (1 + F + KT) * 8 : 0) + ((size_t)KT + M + 1) * 4 +
                       ((size_t)F + M) * PREDICT_THREADS * 4;
  return (bytes + 7) / 8;
}

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
    unsigned long long seed, const uint32_t *__restrict__ row_ids) {
  extern __shared__ double smem[];
  // LDS: [W (C * P doubles) if it fits][keys KT][kbegin M+1][x tile F x 256][slot tile M x 256]
  //      [per wave: queue of selected rows, QCAP u32]
  const int P = 1 + F + KT;
  double *w_l = smem;
  int32_t *keys_l = reinterpret_cast<int32_t *>(smem + (w_in_lds ? (size_t)C * P : 0));
  int32_t *kb_l = keys_l + KT;
  float *x_l = reinterpret_cast<float *>(kb_l + M + 1);
  int32_t *s_l = reinterpret_cast<int32_t *>(x_l + (size_t)F * PREDICT_THREADS);
  // 8-byte aligned: everything before it is a multiple of 8 bytes or padded to one below
  unsigned long long *queue =
      reinterpret_cast<unsigned long long *>(smem + predict_queue_offset(F, M, C, KT, w_in_lds)) +
      (threadIdx.x >> 6) * PREDICT_QCAP;
  const int tid = threadIdx.x, lane = tid & 63;
  if (w_in_lds)
    for (int i = tid; i < C * P; i += PREDICT_THREADS) w_l[i] = W[i];
  for (int i = tid; i < KT; i += PREDICT_THREADS) keys_l[i] = keys[i];
  for (int i = tid; i <= M; i += PREDICT_THREADS) kb_l[i] = kbegin[i];
  __syncthreads();
  const double *w = w_in_lds ? w_l : W;

  auto predict_row = [&](uint64_t row) {
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
        // (row_ids: the rows have been reordered — a row keeps the draw of its original place)
        const unsigned long long rid = row_ids ? (unsigned long long)row_ids[row] : (unsigned long long)row;
        const unsigned long long h = mix64(seed + 0x9E3779B97F4A7C15ull * (rid + 1));
        const double u1 = ((double)(h >> 32) + 1.0) * (1.0 / 4294967296.0);  // (0, 1]
        const double u2 = (double)(h & 0xFFFFFFFFull) * (1.0 / 4294967296.0);
        best += noise_sd * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
      }
      out_f[row] = (float)best;
    }
  };

  const uint64_t wave = ((uint64_t)blockIdx.x * PREDICT_THREADS + tid) >> 6;
  const uint64_t nwaves = ((uint64_t)gridDim.x * PREDICT_THREADS) >> 6;
  if (!mask) {
    for (uint64_t base = wave * 64; base < rows; base += nwaves * 64)
      if (base + lane < rows) predict_row(base + lane);
    return;
  }
  // Row filter: a wave scans 256 mask bytes at a time, appends the selected rows to its queue
  // (ballot + prefix popcount) and predicts them 64 at a time, so that a sparse filter (the 10 %
  // missing values of one column) still keeps every lane busy.
  const bool aligned = (reinterpret_cast<uintptr_t>(mask) & 3) == 0;
  unsigned qn = 0;                                    // queue fill, the same in every lane
  for (uint64_t base = wave * 256; base < rows; base += nwaves * 256) {
    const uint64_t r0 = base + 4 * (uint64_t)lane;
    unsigned mbits = 0;
    if (aligned && r0 + 4 <= rows) mbits = *reinterpret_cast<const unsigned *>(mask + r0);
    else
      for (int j = 0; j < 4; j++)
        if (r0 + j < rows) mbits |= (unsigned)mask[r0 + j] << (8 * j);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const bool sel = ((mbits >> (8 * j)) & 0xFFu) != 0;
      const unsigned long long b = __ballot(sel);
      if (sel) queue[qn + __popcll(b & ((1ull << lane) - 1))] = r0 + j;
      qn += (unsigned)__popcll(b);
    }
    while (qn >= 64) {                                // the top 64 entries, one per lane
      predict_row(queue[qn - 64 + lane]);
      qn -= 64;
    }
  }
  if ((unsigned)lane < qn) predict_row(queue[lane]);
}

}  // namespace

size_t predict_lds_bytes(int F, int M, int C, int KT, size_t lds_limit, int *w_in_lds) {
  const size_t qbytes = (size_t)(PREDICT_THREADS / 64) * PREDICT_QCAP * 8;
  *w_in_lds = predict_queue_offset(F, M, C, KT, 1) * 8 + qbytes <= lds_limit;
  return predict_queue_offset(F, M, C, KT, *w_in_lds) * 8 + qbytes;
}

hipError_t launch_predict(bool argmax, const NumCols &num, const CatCols &cat, int F, int M, int C,
                          int KT, const int32_t *kbegin, const int32_t *keys, const double *W,
                          const uint8_t *mask, uint64_t rows, float *out_f, int32_t *out_i,
                          const int32_t *labels, int noise, double noise_sd,
                          unsigned long long seed, int grid, size_t lds_limit, hipStream_t stream,
                          const uint32_t *row_ids) {
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
        noise_sd, seed, row_ids);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&predict_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    predict_kernel<false><<<grid, PREDICT_THREADS, lds, stream>>>(
        num, cat, F, M, C, KT, kbegin, keys, W, w_in_lds, mask, rows, out_f, out_i, labels, noise,
        noise_sd, seed, row_ids);
  }
  return hipGetLastError();
}

}  // namespace cofactor
