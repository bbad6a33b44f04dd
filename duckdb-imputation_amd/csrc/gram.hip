// Dense part of sum_to_triple_n_m on gfx950: lin_agg (column sums) and quad_agg (upper triangle of
// X^T X) of n <= 20 float columns, streamed once from HBM.
//
// Replaces the reference's n(n+1)/2 passes over each DataChunk
// (duckdb_extension/src/triple/sum/sum_no_lift.cpp:119-146).
//
// Shape of the kernel (DESIGN.md "gram_kernel"):
//   * a 256-thread workgroup walks 256-row tiles of the table, grid-stride;
//   * each tile's n columns are fetched with coalesced 16-B/lane global loads (one wave-instruction
//     = 1 KiB of ONE column) one tile ahead of use, and parked in LDS column-major
//     [column][row] with a +4-float column stride;
//   * each wave owns 64 rows of the tile; per 4 rows every lane reads its A and B operand columns
//     with two ds_read_b128 and issues 4 v_mfma_f32_4x4x1_16b_f32 — one MFMA per row covers all
//     <= 15 4x4 blocks of the upper triangle (device.hpp); for n <= 12 the spare blocks of the
//     instruction take further rows (16 / 4 / 2 rows per MFMA at n <= 4 / 8 / 12);
//   * narrow tables keep several tiles' loads in flight (a tile is only n KiB): a register ring
//     of 5 / 3 / 2 tiles at n <= 4 / 8 / 20, with unconditional loads so the waits are vmcnt(k);
//   * fp32 MFMA chains are cut every FLUSH_TILES tiles and folded into fp64 registers (the
//     reference's fp32 running sums saturate at 2^24);
//   * per-workgroup fp64 images go to a [slot][workgroup] scratch array and a second small kernel
//     adds them, in a fixed order, into the aggregate's accumulator image.
#include "device.hpp"

#include <algorithm>
#include <cstdlib>

namespace cofactor {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));


namespace {

constexpr int FLUSH_TILES = 4;  // 4 tiles x 16 rows per chain = 64 fp32 adds between fp64 folds

template <bool ALIGNED>
__device__ __forceinline__ float4 load_rows4_full(const float *__restrict__ col, uint64_t row) {
  // every byte is read exactly once: non-temporal loads keep the stream from churning L2 / MALL
  if (ALIGNED) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(col + row));
    return make_float4(v[0], v[1], v[2], v[3]);
  }
  return make_float4(__builtin_nontemporal_load(col + row), __builtin_nontemporal_load(col + row + 1),
                     __builtin_nontemporal_load(col + row + 2), __builtin_nontemporal_load(col + row + 3));
}

// last, partial tile only: rows past the end contribute zeros (neutral for every sum)
__device__ __forceinline__ float4 load_rows4_tail(const float *__restrict__ col, uint64_t row,
                                                  uint64_t rows) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < rows) v.x = col[row];
  if (row + 1 < rows) v.y = col[row + 1];
  if (row + 2 < rows) v.z = col[row + 2];
  if (row + 3 < rows) v.w = col[row + 3];
  return v;
}

#ifdef COFACTOR_DEV_ABLATE
__device__ int g_gram_ablate = 0;   // timing experiments only: 1 = no MFMA loop, 2 = loads only
#endif

// Rows one MFMA handles: the instruction has 16 independent 4x4 blocks and the upper triangle of
// NB column blocks needs NPAIR of them, so for n <= 12 the spare blocks take further ROWS (block
// b serves block pair b % NPAIR of row b / NPAIR); their accumulators are added at the end.
__host__ __device__ constexpr int gram_rows_per_mfma(int n) {
  return n <= 4 ? 16 : (n <= 8 ? 4 : (n <= 12 ? 2 : 1));
}
// Tiles whose loads a thread keeps in flight: a 256-row tile is only n KiB, so narrow tables need
// a deeper ring to cover the HBM latency (about 80 KiB in flight per CU at 4 workgroups).
__host__ __device__ constexpr int gram_ring_depth(int n) {
  return n <= 4 ? 5 : (n <= 8 ? 3 : 2);
}

template <int N, bool ALIGNED, bool MASKED>
__global__ __launch_bounds__(GRAM_THREADS) void gram_kernel(NumCols cols, uint64_t rows,
                                                            double *__restrict__ partials,
                                                            const uint8_t *__restrict__ mask_arg) {
  const uint8_t *__restrict__ mask = MASKED ? mask_arg : nullptr;   // unfiltered variant: no filter bytes are read
  constexpr int NB = (N + 3) / 4;
  constexpr int NPAIR = NB * (NB + 1) / 2;
  constexpr int NBC = 4 * NB;                     // data columns incl. zero padding to 4*NB
  constexpr int CS = GRAM_COL_STRIDE;
  constexpr int LD = (N * 64 + GRAM_THREADS - 1) / GRAM_THREADS;  // float4 loads / thread / tile
  constexpr int RPM = gram_rows_per_mfma(N);
  constexpr int DEPTH = gram_ring_depth(N);
  // columns [0,N) data, [N,NBC) zero padding, column NBC all-zero (operand of unused blocks)
  constexpr int TILE_FLOATS = (NBC + 1) * CS > 8 * GRAM_ACC_LEN ? (NBC + 1) * CS : 8 * GRAM_ACC_LEN;
  __shared__ __attribute__((aligned(16))) float tile[TILE_FLOATS];  // also the wave-fold scratch

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int colA = NBC, colB = NBC, rsub = 0;
  {
    const int b = lane >> 2, t = lane & 3;
    if (b < RPM * NPAIR) {
      rsub = b / NPAIR;
      int bi = 0, rem = b % NPAIR;
      while (rem >= NB - bi) { rem -= NB - bi; bi++; }
      colA = 4 * bi + t;
      colB = 4 * (bi + rem) + t;
    }
  }
  for (int i = tid; i < (NBC + 1 - N) * CS; i += GRAM_THREADS) tile[N * CS + i] = 0.f;

  // Whole tiles go through the register ring: their loads are unconditional (a ring slot past
  // the end re-reads the last whole tile), so the compiler waits with vmcnt(k) for the oldest
  // slot only.  The partial last tile is handled once, after the loop, by the workgroup it
  // falls to.
  const uint64_t nfull = rows / GRAM_TILE_ROWS;
  // (the row filter is 4-byte aligned here: launch_gram peels the rows before the first aligned
  // byte off into a launch of their own)
  auto fetch = [&](float4 (&pre)[LD], unsigned &pre_mask, uint64_t t) {
    const uint64_t r0 = t * GRAM_TILE_ROWS + 4 * (uint64_t)lane;
    // raw word; park() looks at it.  (Looking at it here would make the wave wait for this load
    // right away, and with it for every older load of the ring.)
    pre_mask = MASKED ? *reinterpret_cast<const unsigned *>(mask + r0) : 0x01010101u;
#pragma unroll
    for (int i = 0; i < LD; i++) {
      const int col = min(wave + 4 * i, N - 1);   // wave-uniform: q = tid + 256 i, col = q / 64
      pre[i] = load_rows4_full<ALIGNED>(cols.p[col], r0);
    }
  };
  auto fetch_tail = [&](float4 (&pre)[LD], unsigned &pre_mask, uint64_t t) {
    const uint64_t r0 = t * GRAM_TILE_ROWS + 4 * (uint64_t)lane;
    pre_mask = 0x01010101u;
    if (mask) {
      pre_mask = 0u;
      for (int e = 0; e < 4; e++)
        if (r0 + e < rows) pre_mask |= (unsigned)mask[r0 + e] << (8 * e);
    }
#pragma unroll
    for (int i = 0; i < LD; i++) {
      const int col = min(wave + 4 * i, N - 1);
      pre[i] = load_rows4_tail(cols.p[col], r0, rows);
    }
  };
  auto park = [&](const float4 (&pre)[LD], unsigned pre_mask) {
#pragma unroll
    for (int i = 0; i < LD; i++) {
      const int col = wave + 4 * i;
      if (4 * i + 3 < N || col < N) {
        float4 v = pre[i];
        if (MASKED) {                             // filtered rows contribute nothing
          v.x = (pre_mask & 0x000000FFu) ? v.x : 0.f;
          v.y = (pre_mask & 0x0000FF00u) ? v.y : 0.f;
          v.z = (pre_mask & 0x00FF0000u) ? v.z : 0.f;
          v.w = (pre_mask & 0xFF000000u) ? v.w : 0.f;
        }
        *reinterpret_cast<float4 *>(&tile[col * CS + 4 * lane]) = v;
      }
    }
  };

  f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  f32x2 ls_lo = {0.f, 0.f}, ls_hi = {0.f, 0.f};         // per-lane column sums of the A operand
  double dq0 = 0, dq1 = 0, dq2 = 0, dq3 = 0, dl = 0;
  // this wave's 64 rows of the tile; a lane serving row group rsub starts 4 * rsub rows in and
  // steps over the RPM groups of each round
  const float *pa = tile + colA * CS + wave * 64 + 4 * rsub;
  const float *pb = tile + colB * CS + wave * 64 + 4 * rsub;
  int since_flush = 0;
  auto crunch = [&]() {
    const f32x4 *va = reinterpret_cast<const f32x4 *>(pa);
    const f32x4 *vb = reinterpret_cast<const f32x4 *>(pb);
#pragma unroll 4
    for (int it = 0; it < 16 / RPM; it++) {
      const f32x4 a = va[it * RPM], b = vb[it * RPM];
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], b[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], b[1], acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], b[2], acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], b[3], acc3, 0, 0, 0);
      ls_lo += __builtin_shufflevector(a, a, 0, 1);       // v_pk_add_f32 on the aligned halves
      ls_hi += __builtin_shufflevector(a, a, 2, 3);
    }
    if (++since_flush == FLUSH_TILES * RPM) {             // still <= 64 fp32 adds per chain
      dq0 += (double)((acc0[0] + acc1[0]) + (acc2[0] + acc3[0]));
      dq1 += (double)((acc0[1] + acc1[1]) + (acc2[1] + acc3[1]));
      dq2 += (double)((acc0[2] + acc1[2]) + (acc2[2] + acc3[2]));
      dq3 += (double)((acc0[3] + acc1[3]) + (acc2[3] + acc3[3]));
      dl += (double)((ls_lo[0] + ls_lo[1]) + (ls_hi[0] + ls_hi[1]));
      acc0 = acc1 = acc2 = acc3 = f32x4{0, 0, 0, 0};
      ls_lo = ls_hi = f32x2{0.f, 0.f};
      since_flush = 0;
    }
  };

#ifdef COFACTOR_DEV_ABLATE
  const int ablate = g_gram_ablate;
#endif
  const uint64_t G = gridDim.x;
  uint64_t t = blockIdx.x;
  if (t < nfull) {
    float4 pre[DEPTH][LD];
    unsigned pmask[DEPTH];
#pragma unroll
    for (int r = 0; r < DEPTH; r++) fetch(pre[r], pmask[r], min(t + r * G, nfull - 1));
    while (t < nfull) {
#pragma unroll
      for (int r = 0; r < DEPTH; r++) {           // tile t sits in ring slot r
        park(pre[r], pmask[r]);
        __syncthreads();
        fetch(pre[r], pmask[r], min(t + DEPTH * G, nfull - 1));   // flies under DEPTH tiles of MFMAs
#ifdef COFACTOR_DEV_ABLATE
        if (ablate == 0)
#endif
        crunch();
        __syncthreads();                          // everyone done reading before the next park()
        t += G;
        if (t >= nfull) break;
      }
    }
  }
  if (rows % GRAM_TILE_ROWS && blockIdx.x == nfull % G) {   // the partial last tile
    float4 pre[LD];
    unsigned pmask;
    fetch_tail(pre, pmask, nfull);
    park(pre, pmask);
    __syncthreads();
    crunch();
    __syncthreads();
  }
  dq0 += (double)((acc0[0] + acc1[0]) + (acc2[0] + acc3[0]));
  dq1 += (double)((acc0[1] + acc1[1]) + (acc2[1] + acc3[1]));
  dq2 += (double)((acc0[2] + acc1[2]) + (acc2[2] + acc3[2]));
  dq3 += (double)((acc0[3] + acc1[3]) + (acc2[3] + acc3[3]));
  dl += (double)((ls_lo[0] + ls_lo[1]) + (ls_hi[0] + ls_hi[1]));

  // fold the 4 waves and the RPM row groups (fixed order) through LDS: one image per workgroup,
  // in the layout of device.hpp (lane 4 * pair + t)
  double *red = reinterpret_cast<double *>(tile);  // 4 waves x 320 doubles = 10 KiB
  static_assert(sizeof(double) * 4 * GRAM_ACC_LEN <= sizeof(float) * TILE_FLOATS,
                "wave fold scratch must fit the tile");
  __syncthreads();
  double *mine = red + wave * GRAM_ACC_LEN;
  mine[0 * 64 + lane] = dq0;
  mine[1 * 64 + lane] = dq1;
  mine[2 * 64 + lane] = dq2;
  mine[3 * 64 + lane] = dq3;
  mine[4 * 64 + lane] = dl;
  __syncthreads();
  for (int i = tid; i < GRAM_ACC_LEN; i += GRAM_THREADS) {
    const int ln = i & 63;
    double v = 0;
    if (ln < 4 * NPAIR)
#pragma unroll
      for (int rs = 0; rs < RPM; rs++) {
        const int j = i + 4 * NPAIR * rs;
        v += ((red[j] + red[GRAM_ACC_LEN + j]) + red[2 * GRAM_ACC_LEN + j]) + red[3 * GRAM_ACC_LEN + j];
      }
    partials[(uint64_t)i * gridDim.x + blockIdx.x] = v;
  }
}

// acc[i] += sum over workgroups of partials[i][wg], fixed tree order: bitwise reproducible.
__global__ __launch_bounds__(256) void gram_fold_kernel(const double *__restrict__ partials,
                                                        int nwg, double *__restrict__ acc) {
  __shared__ double red[256];
  const int i = blockIdx.x;
  double v = 0;
  for (int w = threadIdx.x; w < nwg; w += 256) v += partials[(uint64_t)i * nwg + w];
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) acc[i] += red[0];
}

template <int N>
hipError_t launch_n(const NumCols &cols, uint64_t rows, int grid, double *partials,
                    const uint8_t *mask, hipStream_t stream) {
  bool aligned = true;
  for (int k = 0; k < N; k++) aligned = aligned && ((reinterpret_cast<uintptr_t>(cols.p[k]) & 15) == 0);
  if (aligned && mask)
    hipLaunchKernelGGL((gram_kernel<N, true, true>), dim3(grid), dim3(GRAM_THREADS), 0, stream, cols, rows, partials, mask);
  else if (aligned)
    hipLaunchKernelGGL((gram_kernel<N, true, false>), dim3(grid), dim3(GRAM_THREADS), 0, stream, cols, rows, partials, mask);
  else if (mask)
    hipLaunchKernelGGL((gram_kernel<N, false, true>), dim3(grid), dim3(GRAM_THREADS), 0, stream, cols, rows, partials, mask);
  else
    hipLaunchKernelGGL((gram_kernel<N, false, false>), dim3(grid), dim3(GRAM_THREADS), 0, stream, cols, rows, partials, mask);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_gram_fold(const double *partials, int nwg, double *acc, hipStream_t stream) {
  hipLaunchKernelGGL(gram_fold_kernel, dim3(GRAM_ACC_LEN), dim3(256), 0, stream, partials, nwg, acc);
  return hipGetLastError();
}

// counter += number of rows whose mask byte is non-zero
__global__ __launch_bounds__(256) void count_mask_kernel(const uint8_t *__restrict__ mask, uint64_t rows,
                                                         unsigned long long *__restrict__ counter) {
  unsigned long long kept = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) kept += mask[r] != 0;
  for (int off = 32; off > 0; off >>= 1) kept += __shfl_down(kept, off, 64);
  if ((threadIdx.x & 63) == 0 && kept) atomicAdd(counter, kept);
}

hipError_t launch_count_mask(const uint8_t *mask, uint64_t rows, unsigned long long *counter, hipStream_t stream) {
  if (rows == 0) return hipSuccess;
  uint64_t blocks = (rows + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(count_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, mask, rows, counter);
  return hipGetLastError();
}

// kernel for n columns over `rows` rows (filter, if any, 4-byte aligned unless rows < one tile)
static hipError_t launch_gram_kernel(const NumCols &cols, int n, uint64_t rows, int grid, double *partials,
                                     const uint8_t *mask, hipStream_t stream) {
  hipError_t e = hipErrorInvalidValue;
  switch (n) {
#define CASE(N) case N: e = launch_n<N>(cols, rows, grid, partials, mask, stream); break;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10)
    CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18) CASE(19) CASE(20)
#undef CASE
    default: return hipErrorInvalidValue;
  }
  return e;
}

// kernel + fold of the per-workgroup images into acc
static hipError_t launch_gram_rows(const NumCols &cols, int n, uint64_t rows, int grid, double *partials,
                                   double *acc, hipStream_t stream, const uint8_t *mask) {
  if (rows == 0) return hipSuccess;
  const uint64_t ntiles = (rows + GRAM_TILE_ROWS - 1) / GRAM_TILE_ROWS;
  if ((uint64_t)grid > ntiles) grid = (int)ntiles;
  hipError_t e = launch_gram_kernel(cols, n, rows, grid, partials, mask, stream);
  if (e != hipSuccess) return e;
  return launch_gram_fold(partials, grid, acc, stream);
}

hipError_t launch_gram(const NumCols &cols, int n, uint64_t rows, int grid, double *partials,
                       double *acc, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
                       const uint8_t *mask) {
  if (rows == 0 || n == 0) return hipSuccess;
#ifdef COFACTOR_DEV_ABLATE
  {
    const char *v = getenv("COFACTOR_GRAM_ABLATE");
    int abl = v ? atoi(v) : 0;
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_gram_ablate), &abl, sizeof(int), 0, hipMemcpyHostToDevice, stream);
    (void)hipStreamSynchronize(stream);
  }
#endif
  hipError_t e;
  if (ev0 && (e = hipEventRecord(ev0, stream)) != hipSuccess) return e;
  uint64_t head = 0;
  if (mask && (reinterpret_cast<uintptr_t>(mask) & 3)) {
    // the kernel reads the filter of whole tiles as 4-byte words: the 1..3 rows before the first
    // aligned byte go into a launch of their own (below one tile: byte-wise tail path)
    head = std::min<uint64_t>(rows, 4 - (reinterpret_cast<uintptr_t>(mask) & 3));
    if ((e = launch_gram_rows(cols, n, head, grid, partials, acc, stream, mask)) != hipSuccess) return e;
  }
  NumCols rest = cols;
  for (int k = 0; k < n; k++) rest.p[k] = cols.p[k] + head;
  const uint64_t rrows = rows - head;
  if (rrows == 0) return ev1 ? hipEventRecord(ev1, stream) : hipSuccess;
  const uint64_t ntiles = (rrows + GRAM_TILE_ROWS - 1) / GRAM_TILE_ROWS;
  if ((uint64_t)grid > ntiles) grid = (int)ntiles;
  if ((e = launch_gram_kernel(rest, n, rrows, grid, partials, mask ? mask + head : nullptr, stream)) != hipSuccess) return e;
  if (ev1 && (e = hipEventRecord(ev1, stream)) != hipSuccess) return e;
  return launch_gram_fold(partials, grid, acc, stream);
}

// ---- dense seam of the multi-GPU path (SumStateCombine across ranks, sum_state.cpp:25,73-83) ----
// out[0] = N, out[1..n] = lin, then quad (row-major upper triangle, or the diagonal for NB): the
// state's dense totals straight from its accumulator image, no host round trip.
__global__ __launch_bounds__(256) void dense_export_kernel(const double *__restrict__ acc,
                                                           const unsigned long long *__restrict__ kept,
                                                           double n_base, const double *__restrict__ extra,
                                                           int n, int kind, double *__restrict__ out) {
  const int len = 1 + n + (kind ? n : n * (n + 1) / 2);
  const int i = threadIdx.x;
  if (i >= len) return;
  double v;
  if (i == 0) v = n_base + (double)*kept;
  else if (i <= n) v = acc[gram_lin_pos(i - 1, n)];
  else if (kind) v = acc[gram_quad_pos(i - 1 - n, i - 1 - n, n)];
  else {
    int q = i - 1 - n, j = 0;
    while (q >= n - j) { q -= n - j; j++; }
    v = acc[gram_quad_pos(j, j + q, n)];
  }
  if (extra) v += extra[i];                       // what the state holds on the host (combine, lifted triples)
  out[i] = v;
}

// The reverse: the accumulator image becomes exactly the totals in `in` (after the all-reduce).
__global__ __launch_bounds__(GRAM_ACC_LEN) void dense_import_kernel(const double *__restrict__ in, int n,
                                                                    int kind, double *__restrict__ acc,
                                                                    unsigned long long *__restrict__ kept) {
  __shared__ double img[GRAM_ACC_LEN];
  const int len = 1 + n + (kind ? n : n * (n + 1) / 2);
  const int i = threadIdx.x;
  img[i] = 0.0;
  __syncthreads();
  if (i == 0) *kept = (unsigned long long)(in[0] + 0.5);
  else if (i <= n) img[gram_lin_pos(i - 1, n)] = in[i];
  else if (i < len) {
    if (kind) img[gram_quad_pos(i - 1 - n, i - 1 - n, n)] = in[i];
    else {
      int q = i - 1 - n, j = 0;
      while (q >= n - j) { q -= n - j; j++; }
      img[gram_quad_pos(j, j + q, n)] = in[i];
    }
  }
  __syncthreads();
  acc[i] = img[i];
}

hipError_t launch_dense_export(const double *acc, const unsigned long long *kept, double n_base,
                               const double *extra, int n, int kind, double *out, hipStream_t stream) {
  hipLaunchKernelGGL(dense_export_kernel, dim3(1), dim3(256), 0, stream, acc, kept, n_base, extra, n, kind, out);
  return hipGetLastError();
}

hipError_t launch_dense_import(const double *in, int n, int kind, double *acc, unsigned long long *kept,
                               hipStream_t stream) {
  hipLaunchKernelGGL(dense_import_kernel, dim3(1), dim3(GRAM_ACC_LEN), 0, stream, in, n, kind, acc, kept);
  return hipGetLastError();
}

// ---- calibration: what a plain streaming kernel reaches on this GPU (bench.py's second roofline) ---
// float4 per lane, grid-stride, non-temporal, the access shape of gram_kernel's fetch.
__global__ __launch_bounds__(256) void calib_copy_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst,
                                                         uint64_t n4) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
    const f32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
    __builtin_nontemporal_store(a, dst + i);
    __builtin_nontemporal_store(b, dst + i + stride);
    __builtin_nontemporal_store(c, dst + i + 2 * stride);
    __builtin_nontemporal_store(d, dst + i + 3 * stride);
  }
  for (; i < n4; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
__global__ __launch_bounds__(256) void calib_read_kernel(const f32x4 *__restrict__ src, float *__restrict__ out,
                                                         uint64_t n4) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  f32x4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    s0 += __builtin_nontemporal_load(src + i);
    s1 += __builtin_nontemporal_load(src + i + stride);
    s2 += __builtin_nontemporal_load(src + i + 2 * stride);
    s3 += __builtin_nontemporal_load(src + i + 3 * stride);
  }
  for (; i < n4; i += stride) s0 += __builtin_nontemporal_load(src + i);
  const f32x4 s = (s0 + s1) + (s2 + s3);
  const float v = (s[0] + s[1]) + (s[2] + s[3]);
  if (v == 12345.678f) out[0] = v;                 // keeps the loads alive; practically never true
}

hipError_t launch_calibration(const void *src, void *dst, uint64_t bytes, int grid, bool copy, hipStream_t stream) {
  const uint64_t n4 = bytes / 16;
  if (copy)
    hipLaunchKernelGGL(calib_copy_kernel, dim3(grid), dim3(256), 0, stream, (const f32x4 *)src, (f32x4 *)dst, n4);
  else
    hipLaunchKernelGGL(calib_read_kernel, dim3(grid), dim3(256), 0, stream, (const f32x4 *)src, (float *)dst, n4);
  return hipGetLastError();
}

}  // namespace cofactor
