// Per-key counts and per-key sums of the numeric columns for key columns of up to 64 keys, on the
// matrix cores — the code-cache route's cat_sums pass (cat.hip) without its n x m ds_add_f64 per row
// (17 .. 64 keys per column at n <= 10; <= 32 at n <= 20; the wide shapes, m > 10, at <= 16 keys).
//
// Replaces, for those columns, the per-row map updates of Triple::SumNoLift
// (duckdb_extension/src/triple/sum/sum_no_lift.cpp:157-193: lin_cat[col][key] += 1,
// quad_num_cat[col][key][j] += x_j).
//
// For every key column c and every block of 16 codes: S_c[code][j] = sum over rows of
// onehot(code_c(row) == code) * x_j(row) is a matrix product, v_mfma_f32_16x16x32_bf16 with
//   A = the one-hot of 32 rows against the block's 16 codes (bf16 1.0: three packed VALU operations per
//       two rows from the 16-bit codes of the cache),
//   B = the rows' numeric values split into three bf16 pieces (x = hi + mid + lo, exact), one column
//       per piece and numeric column, plus a column of ones for the counts,
// so 64 keys are four blocks per column: 4 x 2 MFMAs per column and 32 rows at n = 10.
// A 256-thread workgroup walks 64-row tiles: all threads split the tile's values into pieces (LDS,
// B-operand order), then every wave takes its share of the key columns (wave w: the w-th, (w+4)-th,
// (w+8)-th selected column) through all their code blocks.  The next tile's values and codes are
// loaded while this one is multiplied.  fp32 accumulators are folded every 32 tiles into an fp64
// table in LDS (as fused.hip does: <= 2048 adds of bf16 pieces per chain); the table goes to the
// aggregate's tables once per workgroup.
// Rows the filter dropped and rows past the end carry CODE_NONE in the cache and match no code.
#include "device.hpp"

namespace cofactor {

typedef __bf16 cs_bf16x8 __attribute__((ext_vector_type(8)));
typedef float cs_f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int CS_TILE = 64;
constexpr int CS_PST = 144;           // bytes of one piece column in LDS: 64 rows bf16 + 16 (bank spread)
constexpr int CS_FOLD_TILES = 32;
constexpr int CS_MC = 3;              // key columns per wave
// CS_TW threads (template parameter): four waves share a tile's key columns, eight when there are more
// than twelve of them (wide shapes: one split of the numeric values into pieces serves all key columns)

// 8 u16 codes (two uint2) -> 8 bf16 one-hot values for code `ii` (in both halves): d = code ^ ii is 0
// only on a match; min(d, 1) is 0 / 1; 0x3F80 + min * 0xC080 (mod 2^16) is bf16 1.0 or 0.
__device__ __forceinline__ cs_bf16x8 cs_onehot8(uint2 lo, uint2 hi, unsigned ii) {
  unsigned w[4] = {lo.x, lo.y, hi.x, hi.y};
  const unsigned ones = 0x00010001u, neg = 0xC080C080u, one_bf = 0x3F803F80u;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    unsigned m, r;
    const unsigned d = w[e] ^ ii;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(d), "v"(ones));
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(neg), "v"(one_bf));
    w[e] = r;
  }
  return __builtin_bit_cast(cs_bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
}

template <int KB, int NBB, int CS_TW>
__global__ __launch_bounds__(CS_TW, CS_TW == 256 ? 2 : 1) void cat_sums_mfma_kernel(NumCols num, const unsigned short *__restrict__ codes, uint64_t rows,
                                                            uint64_t stride, CatLayout L, CatDevice D, unsigned col_mask) {
  constexpr int CS_NW = CS_TW / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int l_sel[COFACTOR_MAX_CAT];
  __shared__ int l_nsel;
  const int n = L.n, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n16 = lane & 15, g = lane >> 4;
  if (tid == 0) {
    int k = 0;
    for (int c = 0; c < L.m; c++) if ((col_mask >> c) & 1u) l_sel[k++] = c;
    l_nsel = k;
  }
  __syncthreads();
  const int nsel = l_nsel, W1 = n + 1;                 // table row: n sums, then the count
  // two piece buffers, [16 NBB piece columns][CS_PST bytes] each: a tile is written into one while the
  // slower waves may still read the other — ONE barrier per tile
  unsigned short *pt0 = reinterpret_cast<unsigned short *>(lds_raw);
  double *l_tab = reinterpret_cast<double *>(lds_raw + 2 * 16 * NBB * CS_PST);    // [selected column][16 KB codes][n + 1]
  for (int i = tid; i < 2 * 16 * NBB * CS_PST / 2; i += CS_TW) pt0[i] = 0;
  for (int i = tid; i < nsel * KB * 16 * W1; i += CS_TW) l_tab[i] = 0.0;
  // (a wave with fewer than CS_MC columns — or none — still issues its loads: from the first selected column)
  const int first_col = __builtin_amdgcn_readfirstlane(l_sel[0]);
  int myc[CS_MC];
#pragma unroll
  for (int ci = 0; ci < CS_MC; ci++) myc[ci] = __builtin_amdgcn_readfirstlane(wave + CS_NW * ci < nsel ? l_sel[wave + CS_NW * ci] : -1);
  cs_f32x4 acc[CS_MC][KB][NBB];
#pragma unroll
  for (int ci = 0; ci < CS_MC; ci++)
#pragma unroll
    for (int kb = 0; kb < KB; kb++)
#pragma unroll
      for (int bb = 0; bb < NBB; bb++) acc[ci][kb][bb] = cs_f32x4{0.f, 0.f, 0.f, 0.f};

  // this thread's share of the values: row r of the tile, numeric columns j0, j0 + CS_NW, ..
  const int r = tid & 63, j0 = tid >> 6;
  constexpr int XQ = (5 * NBB + 1 + CS_NW - 1) / CS_NW;   // numeric columns per thread: 16 NBB piece columns hold <= 5 NBB + 1
  const uint64_t ntiles = (rows + CS_TILE - 1) / CS_TILE;
  float xn[XQ];
  uint2 cn[CS_MC][2][2];                              // [column][half of the tile][8 codes as two uint2]
  // Every load is issued unconditionally (addresses clamped, results overridden afterwards): a load
  // inside a branch makes the number of loads in flight unknown to the compiler, which then waits
  // for ALL of them (s_waitcnt vmcnt(0)) before the first use — the prefetch of the next tile included.
  auto fetch = [&](uint64_t tile) {
    const uint64_t row0 = tile * CS_TILE;
    const int rem = (int)min<uint64_t>(rows - row0, (uint64_t)CS_TILE);   // rows of this tile (wave-uniform)
    const uint64_t xr = row0 + (r < rem ? r : 0);
#pragma unroll
    for (int q = 0; q < XQ; q++) {
      const int j = min(j0 + CS_NW * q, n - 1);
      xn[q] = __builtin_nontemporal_load(num.p[j] + xr);
    }
#pragma unroll
    for (int ci = 0; ci < CS_MC; ci++)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        // (a column of the cache is whole 64-row tiles, CODE_NONE behind the last row: no clamping here)
        const unsigned off = 32u * h + 8u * g;
        const unsigned short *col = codes + (uint64_t)(myc[ci] >= 0 ? myc[ci] : first_col) * stride + row0;
        const uint4 v = *reinterpret_cast<const uint4 *>(col + off);
        cn[ci][h][0] = make_uint2(v.x, v.y);
        cn[ci][h][1] = make_uint2(v.z, v.w);
      }
  };
  // what the clamped loads fetched in place of rows past the end / columns this wave does not have
  auto settle = [&](uint64_t tile, float (&xc)[XQ], uint2 (&cc)[CS_MC][2][2]) {
    const uint64_t row0 = tile * CS_TILE;
    const int rem = (int)min<uint64_t>(rows - row0, (uint64_t)CS_TILE);
#pragma unroll
    for (int q = 0; q < XQ; q++) xc[q] = (j0 + CS_NW * q < n && r < rem) ? xn[q] : 0.f;
#pragma unroll
    for (int ci = 0; ci < CS_MC; ci++)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        cc[ci][h][0] = cn[ci][h][0];                   // (a column this wave does not have is skipped below)
        cc[ci][h][1] = cn[ci][h][1];
      }
  };
  auto fold = [&]() {
#pragma unroll
    for (int ci = 0; ci < CS_MC; ci++) {
      if (myc[ci] < 0) continue;
      double *tab = l_tab + (size_t)(wave + CS_NW * ci) * KB * 16 * W1;
#pragma unroll
      for (int kb = 0; kb < KB; kb++)
#pragma unroll
        for (int bb = 0; bb < NBB; bb++) {
          const int pc = 16 * bb + n16;                // piece column: piece * n + numeric column; 3 n: the ones
          const int j = pc < 3 * n ? pc % n : (pc == 3 * n ? n : -1);
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float v = acc[ci][kb][bb][e];
            if (j >= 0 && v != 0.f) unsafeAtomicAdd(&tab[(16 * kb + 4 * g + e) * W1 + j], (double)v);
            acc[ci][kb][bb][e] = 0.f;
          }
        }
    }
  };

  uint64_t tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  int since = 0, buf = 0;
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    unsigned short *pt = pt0 + buf * (16 * NBB * CS_PST / 2);
    // ---- the tile's values into bf16 pieces, B-operand order ----
    const uint64_t row0 = tile * CS_TILE;
    float xc[XQ];
    uint2 cc[CS_MC][2][2];
    settle(tile, xc, cc);
    fetch(min(tile + gridDim.x, ntiles - 1));          // (the last tile once more rather than a branch around the loads)
#pragma unroll
    for (int q = 0; q < XQ; q++) {
      const int j = j0 + CS_NW * q;
      if (j < n) {
        const unsigned u = __float_as_uint(xc[q]);
        unsigned short ph, pm, pl;
        if ((u & 0x7F800000u) == 0x7F800000u) {        // inf / nan (rare): 0 x inf would poison every code's cell
          ph = pm = pl = 0;
          if (row0 + r < rows)
            for (int s = 0; s < nsel; s++) {
              const unsigned code = codes[(uint64_t)l_sel[s] * stride + row0 + r];
              if (code < (unsigned)(16 * KB)) unsafeAtomicAdd(&l_tab[((size_t)s * KB * 16 + code) * W1 + j], (double)xc[q]);
            }
        } else {
          const float r1 = xc[q] - __uint_as_float(u & 0xFFFF0000u);
          const float r2 = r1 - __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
          ph = (unsigned short)(u >> 16); pm = (unsigned short)(__float_as_uint(r1) >> 16); pl = (unsigned short)(__float_as_uint(r2) >> 16);
        }
        pt[(0 * n + j) * (CS_PST / 2) + r] = ph;
        pt[(1 * n + j) * (CS_PST / 2) + r] = pm;
        pt[(2 * n + j) * (CS_PST / 2) + r] = pl;
      }
    }
    if (j0 == 0) pt[(3 * n) * (CS_PST / 2) + r] = 0x3F80;    // the column of ones (counts)
    __syncthreads();
    // ---- every wave: its key columns against the tile ----
    cs_bf16x8 bop[2][NBB];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int bb = 0; bb < NBB; bb++)
        bop[h][bb] = __builtin_bit_cast(cs_bf16x8, *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(pt) + (16 * bb + n16) * CS_PST + (32 * h + 8 * g) * 2));
#pragma unroll
    for (int ci = 0; ci < CS_MC; ci++) {
      if (myc[ci] < 0) continue;
#pragma unroll
      for (int kb = 0; kb < KB; kb++) {
        const unsigned ii = (unsigned)(16 * kb + n16) * 0x00010001u;
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const cs_bf16x8 a = cs_onehot8(cc[ci][h][0], cc[ci][h][1], ii);
#pragma unroll
          for (int bb = 0; bb < NBB; bb++) acc[ci][kb][bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bop[h][bb], acc[ci][kb][bb], 0, 0, 0);
        }
        if (KB * NBB > 4) __builtin_amdgcn_sched_barrier(0);   // one block's one-hots at a time: hoisting them all costs 100 registers
      }
    }
    if (++since == CS_FOLD_TILES) { fold(); since = 0; }
  }
  fold();
  __syncthreads();
  // ---- the workgroup's table into the aggregate's ----
  for (int s = 0; s < nsel; s++) {
    const int c = l_sel[s], kc = L.kc[c];
    const double *tab = l_tab + (size_t)s * KB * 16 * W1;
    for (int i = tid; i < kc * W1; i += CS_TW) {
      const int code = i / W1, j = i - code * W1;
      const double v = tab[i];
      if (v == 0.0) continue;
      if (j == n) atomicAdd(&D.cnt[L.cnt_off[c] + code], (unsigned long long)(v + 0.5));
      else unsafeAtomicAdd(&D.s[L.s_off[c] + (long long)code * n + j], v);
    }
  }
}

}  // namespace

// columns of col_mask: at most 64 codes each, at most 12 of them, triple kind, 1 <= n <= 20, and code
// blocks x piece blocks <= 8 (96 accumulator registers per wave)
bool cat_sums_mfma_applicable(const CatLayout &L, unsigned col_mask, uint64_t rows) {
  if (L.kind != 0 || L.n < 1 || 3 * L.n + 1 > 64 || rows < 4096) return false;
  int nsel = 0, kmax = 0;
  for (int c = 0; c < L.m; c++)
    if ((col_mask >> c) & 1u) { nsel++; kmax = std::max(kmax, L.kc[c]); }
  const int KB = (kmax + 15) / 16, NBB = (3 * L.n + 1 + 15) / 16;
  return nsel >= 1 && nsel <= 8 * CS_MC && kmax >= 1 && kmax <= 64 && KB * NBB <= 8 && (nsel <= 4 * CS_MC || KB == 1);
}

hipError_t launch_cat_sums_mfma(const NumCols &num, const unsigned short *codes, uint64_t rows, uint64_t stride,
                                const CatLayout &L, const CatDevice &D, unsigned col_mask, int wgs, hipStream_t stream) {
  int nsel = 0, kmax = 0;
  for (int c = 0; c < L.m; c++)
    if ((col_mask >> c) & 1u) { nsel++; kmax = std::max(kmax, L.kc[c]); }
  const int KB = (kmax + 15) / 16, NBB = (3 * L.n + 1 + 15) / 16;
  const size_t lds = (size_t)2 * 16 * NBB * CS_PST + (size_t)nsel * KB * 16 * (L.n + 1) * 8;
  const uint64_t ntiles = (rows + CS_TILE - 1) / CS_TILE;
  const int TW = nsel > 4 * CS_MC ? 512 : 256;         // (512: one workgroup per CU)
  const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, (uint64_t)std::max(TW == 512 ? wgs / 2 : wgs, 1));
#define CS_CASE(K, B, T)                                                                                           \
  if (KB == K && NBB == B && TW == T) {                                                                            \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(cat_sums_mfma_kernel<K, B, T>),              \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
    if (e != hipSuccess) return e;                                                                                 \
    hipLaunchKernelGGL((cat_sums_mfma_kernel<K, B, T>), dim3(grid), dim3(T), lds, stream, num, codes, rows, stride, \
                       L, D, col_mask);                                                                            \
    return hipGetLastError();                                                                                      \
  }
  CS_CASE(1, 1, 256) CS_CASE(1, 2, 256) CS_CASE(1, 3, 256) CS_CASE(1, 4, 256) CS_CASE(2, 1, 256) CS_CASE(2, 2, 256)
  CS_CASE(2, 3, 256) CS_CASE(2, 4, 256) CS_CASE(3, 1, 256) CS_CASE(3, 2, 256) CS_CASE(4, 1, 256) CS_CASE(4, 2, 256)
  CS_CASE(1, 1, 512) CS_CASE(1, 2, 512) CS_CASE(1, 3, 512) CS_CASE(1, 4, 512)
#undef CS_CASE
  return hipErrorInvalidValue;
}

}  // namespace cofactor
