// GROUP BY accumulate for wide numeric triples, segmented: the rows of a batch are regrouped by
// group first, then every group's rows go through the matrix cores and reach the group's table row
// once per 2048 rows — instead of one fp64 atomic per cell per row (231 at 20_0) in
// groups_accumulate_kernel (ring.hip).
//
// Replaces, for sum_to_triple_n_0 under GROUP BY, the per-row state pointers of Triple::SumNoLift
// (duckdb_extension/src/triple/sum/sum_no_lift.cpp:84-146): states[sdata.sel->get_index(j)] picks
// the group's state row by row; here the rows are moved next to their group's other rows instead.
//
// Five launches per batch (at most 2^27 rows each):
//   seg_count_kernel    group code of every row (dictionary probe for key-typed groups) + rows per group:
//                       workgroup w owns a contiguous slice of the batch and counts it in an LDS histogram
//                       (no global atomics: 1e8 of them on 1e4 addresses took 5.8 ms), row w of hist
//   seg_prefix_kernel   hist[w][g] -> rows of group g in the slices before w; cnt[g] = the group's rows
//   seg_scan_kernel     exclusive scans: first record of every group, first work unit of every group
//   seg_scatter_kernel  row r -> record [x_0 .. x_{n-1}, 1, 0 ..] (RS = 4 / 8 / 16 / 32 floats, never across a
//                       128-byte line) at the next free place of its group: LDS cursors start at
//                       off[g] + hist[w][g] (same slices as the count), so a group's records follow the slice
//                       order (inside a slice: the order of the LDS atomics); the columns are read coalesced,
//                       a wave parks its 64 records in LDS and RS / 4 neighbouring lanes write one record
//   (more groups than the LDS holds cursors for: global counters and cursors)
//   seg_gram_kernel     one wave per unit (<= SEG_UNIT rows of one group): the unit's records are one run of
//                       memory, fetched with coalesced 16-byte loads, 64 rows at a time through LDS;
//                       X^T X with v_mfma_f32_32x32x2_f32 (RS = 32; lane = column, half-wave = row parity,
//                       the same register is the A and the B operand) or v_mfma_f32_16x16x4_f32 (RS <= 16);
//                       the column of ones yields lin and N; fp32 chains are folded into fp64 every 64 rows
//                       as in gram.hip; the upper triangle goes to the group's row with fp64 atomics
//                       (one per cell per unit: 231 per 2048 rows at 20_0).
// HBM traffic per row at 20_0: 84 B (columns + group id) + 4 + 4 (codes) + 128 written + 128 read = 348 B;
// measured 8.6 ms per 1e8 rows (count 0.96, scatter 5.0, gram 2.5) = 4.0 TB/s, against 154 ms for the
// per-row atomics.
#include "device.hpp"
#include "ring.hpp"

namespace cofactor {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int SEG_UNIT = 2048;      // rows of one group one wave takes

// miss: the word set when a row's group is not in the dictionary (flags[1]: an error the next call
// reports; flags[2]: the optimistic pass of ring_api.cpp, which then inserts the keys and counts again)
__global__ __launch_bounds__(256) void seg_count_atomic_kernel(const int32_t *__restrict__ gid, uint64_t rows, int is_key,
                                                               const unsigned long long *__restrict__ ht_slot,
                                                               const int32_t *__restrict__ ht_code, int ht_cap, long long groups,
                                                               int32_t *__restrict__ code, unsigned *__restrict__ cnt,
                                                               int32_t *__restrict__ miss) {
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (uint64_t)gridDim.x * blockDim.x) {
    int g = gid[r];
    if (is_key) g = cat_lookup_code(ht_slot, ht_code, ht_cap, g);
    if (g < 0 || g >= groups) { *miss = 1; g = -1; }
    else atomicAdd(&cnt[g], 1u);
    code[r] = g;
  }
}

constexpr int SEG_WG = 1024;        // threads of the slice kernels

__device__ __forceinline__ void seg_slice(uint64_t rows, uint64_t &lo, uint64_t &hi) {
  const uint64_t per = ((rows + gridDim.x - 1) / gridDim.x + 3) & ~3ull;
  lo = min(rows, (uint64_t)blockIdx.x * per);
  hi = min(rows, lo + per);
}

__global__ __launch_bounds__(SEG_WG) void seg_count_kernel(const int32_t *__restrict__ gid, uint64_t rows, int is_key,
                                                           const unsigned long long *__restrict__ ht_slot,
                                                           const int32_t *__restrict__ ht_code, int ht_cap, int groups,
                                                           int32_t *__restrict__ code, unsigned *__restrict__ hist,
                                                           int32_t *__restrict__ miss) {
  extern __shared__ unsigned l_hist[];
  for (int g = threadIdx.x; g < groups; g += SEG_WG) l_hist[g] = 0;
  __syncthreads();
  uint64_t lo, hi;
  seg_slice(rows, lo, hi);
  for (uint64_t r = lo + threadIdx.x; r < hi; r += SEG_WG) {
    int g = __builtin_nontemporal_load(gid + r);
    if (is_key) g = cat_lookup_code(ht_slot, ht_code, ht_cap, g);
    if (g < 0 || g >= groups) { *miss = 1; g = -1; }
    else atomicAdd(&l_hist[g], 1u);
    code[r] = g;
  }
  __syncthreads();
  unsigned *mine = hist + (size_t)blockIdx.x * groups;
  for (int g = threadIdx.x; g < groups; g += SEG_WG) mine[g] = l_hist[g];
}

__global__ __launch_bounds__(256) void seg_prefix_kernel(unsigned *__restrict__ hist, int slices, int groups,
                                                         unsigned *__restrict__ cnt) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= groups) return;
  unsigned run = 0;
  for (int w = 0; w < slices; w++) {
    const unsigned c = hist[(size_t)w * groups + g];
    hist[(size_t)w * groups + g] = run;
    run += c;
  }
  cnt[g] = run;
}

// one workgroup: off[g] = rows of the groups before g (off[G] = all), uoff likewise for the units,
// cur[g] = off[g] (the scatter's cursors)
__global__ __launch_bounds__(1024) void seg_scan_kernel(const unsigned *__restrict__ cnt, long long G,
                                                        unsigned *__restrict__ off, unsigned *__restrict__ uoff,
                                                        unsigned *__restrict__ cur) {
  __shared__ unsigned l_r[1024], l_u[1024];
  const int t = threadIdx.x;
  const long long per = (G + 1023) / 1024, lo = min(G, t * per), hi = min(G, lo + per);
  unsigned sr = 0, su = 0;
  for (long long g = lo; g < hi; g++) { const unsigned c = cnt[g]; sr += c; su += (c + SEG_UNIT - 1) / SEG_UNIT; }
  l_r[t] = sr; l_u[t] = su;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const unsigned a = t >= d ? l_r[t - d] : 0, b = t >= d ? l_u[t - d] : 0;
    __syncthreads();
    l_r[t] += a; l_u[t] += b;
    __syncthreads();
  }
  unsigned pr = l_r[t] - sr, pu = l_u[t] - su;
  for (long long g = lo; g < hi; g++) {
    const unsigned c = cnt[g];
    off[g] = pr; cur[g] = pr; uoff[g] = pu;
    pr += c; pu += (c + SEG_UNIT - 1) / SEG_UNIT;
  }
  if (t == 1023) { off[G] = l_r[1023]; uoff[G] = l_u[1023]; }
}

// LDSCUR: the cursors of all groups sit in LDS (started at off[g] + hist[slice][g]); else global
// cursors (cur, started at off[g]) and grid-stride row blocks.
// A wave parks the records of its 64 rows in LDS and writes them out with RS / 4 neighbouring lanes
// per record: a store instruction then covers 64 / (RS / 4) whole records (10 at RS = 24), about 15
// memory requests instead of 64 — with one lane per record the scatter ran at the request rate of
// the L2 (8.3 ms per 1e8 rows at 20_0), not at the rate of the memory.
template <int RS, bool LDSCUR, int TW>
__global__ __launch_bounds__(TW) void seg_scatter_kernel(const int32_t *__restrict__ code, NumCols num, int n, uint64_t rows,
                                                             const unsigned *__restrict__ off, const unsigned *__restrict__ hist,
                                                             int groups, unsigned *__restrict__ cur, float *__restrict__ rec) {
  extern __shared__ __attribute__((aligned(16))) unsigned l_raw[];
  constexpr int R4 = RS / 4, RPI = 64 / R4, NIT = (64 + RPI - 1) / RPI;
  // a parked record takes RS + 4 words: with a stride of RS (a multiple of 32 words at 20_0) the 16-byte
  // writes of a wave's lanes all fall on two bank groups (SQ_LDS_BANK_CONFLICT: 7.1e8 cycles per 1e8 rows)
  constexpr int RSP = RS + 4;
  constexpr int WSTRIDE = 64 * RSP + 64;               // words of one wave's staging: records, then their places
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float *l_rec = reinterpret_cast<float *>(l_raw) + (size_t)wave * WSTRIDE;
  unsigned *l_pos = l_raw + (size_t)wave * WSTRIDE + 64 * RSP;
  unsigned *l_cur = l_raw + (size_t)(TW / 64) * WSTRIDE;
  uint64_t lo, hi, stride;
  if (LDSCUR) {
    const unsigned *mine = hist + (size_t)blockIdx.x * groups;
    for (int g = threadIdx.x; g < groups; g += TW) l_cur[g] = off[g] + mine[g];
    __syncthreads();
    seg_slice(rows, lo, hi);
    stride = TW;
  } else {
    lo = (uint64_t)blockIdx.x * TW; hi = rows; stride = (uint64_t)gridDim.x * TW;
  }
  for (uint64_t r0 = lo + (uint64_t)wave * 64; r0 < hi; r0 += stride) {
    // (issuing the next block's loads before this block's stores, or non-temporal stores: slower)
    const uint64_t r = r0 + lane;
    int g = -1;
    float x[RS];
#pragma unroll
    for (int k = 0; k < RS; k++) x[k] = k == n ? 1.f : 0.f;
    if (r < hi) {
      g = __builtin_nontemporal_load(code + r);
#pragma unroll
      for (int k = 0; k < RS && k < COFACTOR_MAX_NUM; k++)
        if (k < n) x[k] = __builtin_nontemporal_load(num.p[k] + r);
    }
    unsigned pos = 0xFFFFFFFFu;
    if (g >= 0) pos = LDSCUR ? atomicAdd(&l_cur[g], 1u) : atomicAdd(&cur[g], 1u);
    l_pos[lane] = pos;
#pragma unroll
    for (int q = 0; q < R4; q++)
      *reinterpret_cast<f32x4 *>(l_rec + lane * RSP + 4 * q) = f32x4{x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int sub = lane / R4, piece = lane - sub * R4;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int rr = it * RPI + sub;
      if (sub < RPI && rr < 64) {
        const unsigned p = l_pos[rr];
        if (p != 0xFFFFFFFFu)
          *reinterpret_cast<f32x4 *>(rec + (size_t)p * RS + 4 * piece) = *reinterpret_cast<const f32x4 *>(l_rec + rr * RSP + 4 * piece);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

__device__ __forceinline__ void seg_unit_of(const unsigned *__restrict__ uoff, long long G, unsigned u, long long &g) {
  long long lo = 0, hi = G;                          // last g with uoff[g] <= u (groups without rows share their successor's offset)
  while (hi - lo > 1) {
    const long long mid = (lo + hi) >> 1;
    if (uoff[mid] <= u) lo = mid; else hi = mid;
  }
  g = lo;
}

// cell of (i, j), i <= j <= n, in a group's table row: [N | lin | upper triangle, row-major]; j == n is the column of ones
__device__ __forceinline__ int seg_cell(int i, int j, int n) {
  if (j == n) return i == n ? 0 : 1 + i;
  return 1 + n + i * n - (i * (i - 1)) / 2 + (j - i);
}

// One wave per unit.  The unit's records are one contiguous run of memory: the wave fetches it 64 rows
// (RS / 4 coalesced 16-byte loads per lane) at a time, one block ahead, parks the block in its own
// piece of LDS and reads the MFMA operands from there — lane (column j, row h of the instruction's 2
// or 4 rows) takes word [row][j], consecutive lanes consecutive words.  (Operand loads straight from
// memory, 4 bytes per lane and 192 bytes per instruction, reached 3.0 TB/s.)
template <int RS>
__global__ __launch_bounds__(256) void seg_gram_kernel(const float *__restrict__ rec, const unsigned *__restrict__ off,
                                                       const unsigned *__restrict__ uoff, long long G, int n,
                                                       double *__restrict__ tab, long long dtot) {
  constexpr bool W32 = RS > 16;                       // 32x32x2 (two rows per MFMA) or 16x16x4 (four)
  constexpr int R4 = RS / 4, CW = W32 ? 32 : 16, RM = W32 ? 2 : 4, NM = 64 / RM, NA = W32 ? 16 : 4;
  __shared__ __attribute__((aligned(16))) float l_blk[4][64 * RS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & (CW - 1), h = lane / CW;
  float *blk = l_blk[wave];
  const bool col_on = j < RS;
  const unsigned units = uoff[G];
  const unsigned wave0 = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
  for (unsigned u = wave0; u < units; u += nwaves) {
    long long g;
    seg_unit_of(uoff, G, u, g);
    const unsigned seg_lo = off[g], seg_hi = off[g + 1];
    const unsigned start = seg_lo + (u - uoff[g]) * SEG_UNIT, end = min(seg_hi, start + SEG_UNIT);
    typedef float accv __attribute__((ext_vector_type(NA)));
    accv acc;
    double dacc[NA];
#pragma unroll
    for (int r = 0; r < NA; r++) { acc[r] = 0.f; dacc[r] = 0.0; }
    f32x4 nx[R4];
    auto fetch = [&](unsigned base) {                 // 64 records from `base`, zeros past the unit's end
      const f32x4 *src = reinterpret_cast<const f32x4 *>(rec + (size_t)base * RS);
      const unsigned quads = base < end ? (end - base) * R4 : 0;
#pragma unroll
      for (int q = 0; q < R4; q++) {
        const unsigned e = lane + 64 * q;
        nx[q] = e < quads ? __builtin_nontemporal_load(src + e) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    fetch(start);
    for (unsigned base = start; base < end; base += 64) {
#pragma unroll
      for (int q = 0; q < R4; q++) *reinterpret_cast<f32x4 *>(blk + 4 * (lane + 64 * q)) = nx[q];
      fetch(base + 64);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i0 = 0; i0 < NM; i0 += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = col_on ? blk[(RM * (i0 + i) + h) * RS + j] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          if constexpr (W32) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i], v[i], acc, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i], v[i], acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < NA; r++) { dacc[r] += (double)acc[r]; acc[r] = 0.f; }   // 64 fp32 adds per chain (gram.hip's rule)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    double *row = tab + g * dtot;
#pragma unroll
    for (int r = 0; r < NA; r++) {
      const int i = W32 ? 8 * (r >> 2) + 4 * h + (r & 3) : 4 * h + r;
      if (i <= j && j <= n) unsafeAtomicAdd(&row[seg_cell(i, j, n)], dacc[r]);
    }
  }
}

}  // namespace

// A record never straddles a 128-byte line: 16, 32, 64 or 128 bytes.  The memory charges a scattered
// write by the line it touches — 96-byte records at 20_0 (1.75 lines each) took 7.1 ms per 1e8 rows to
// write and 2.2 to read back, 128-byte ones 5.0 and 2.5.
int groups_seg_record_floats(int n) { return n + 1 <= 4 ? 4 : (n + 1 <= 8 ? 8 : (n + 1 <= 16 ? 16 : 32)); }

constexpr long long SEG_LDS_GROUPS = 12288;          // groups whose cursors fit the LDS next to the scatter's staged records

static int seg_slices(int cus, long long) { return cus; }   // one 1024-thread workgroup per CU (the scatter's LDS)

size_t groups_seg_scratch_bytes(int n, uint64_t rows, long long groups, int cus) {
  const size_t G = (size_t)groups + 1;
  const size_t hist = groups <= SEG_LDS_GROUPS ? (size_t)seg_slices(cus, groups) * (size_t)groups * 4 : 0;
  return (size_t)rows * 4 + 4 * G * 4 + hist + 512 + (size_t)rows * groups_seg_record_floats(n) * 4;
}

hipError_t launch_groups_segmented(const int32_t *gid, const NumCols &num, int n, uint64_t rows, const CatLayout &Lg,
                                   const CatDevice &Dg, int is_key, long long groups, double *tab, long long dtot,
                                   void *scratch, int cus, int32_t *miss, int phase, hipStream_t stream) {
  const size_t G = (size_t)groups + 1;
  const bool lds = groups <= SEG_LDS_GROUPS;
  const int slices = seg_slices(cus, groups);
  char *p = reinterpret_cast<char *>(scratch);
  int32_t *code = reinterpret_cast<int32_t *>(p); p += (size_t)rows * 4;
  unsigned *cnt = reinterpret_cast<unsigned *>(p); p += G * 4;
  unsigned *off = reinterpret_cast<unsigned *>(p); p += G * 4;
  unsigned *uoff = reinterpret_cast<unsigned *>(p); p += G * 4;
  unsigned *cur = reinterpret_cast<unsigned *>(p); p += G * 4;
  unsigned *hist = reinterpret_cast<unsigned *>(p); p += lds ? (size_t)slices * (size_t)groups * 4 : 0;
  p = reinterpret_cast<char *>(((uintptr_t)p + 255) & ~(uintptr_t)255);
  float *rec = reinterpret_cast<float *>(p);
  const unsigned grid = (unsigned)std::min<uint64_t>((rows + 255) / 256, (uint64_t)cus * 16);
  const size_t lds_bytes = (size_t)groups * 4;
  if (phase != 1) {                                  // phase 0: everything; 1: after the count; 2: the count only
    if (lds) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(seg_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SEG_LDS_GROUPS * 4));
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(seg_count_kernel, dim3(slices), dim3(SEG_WG), lds_bytes, stream, gid, rows, is_key, Dg.ht_slot, Dg.ht_code,
                         Lg.ht_cap[0], (int)groups, code, hist, miss);
    } else {
      hipError_t e = hipMemsetAsync(cnt, 0, G * 4, stream);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(seg_count_atomic_kernel, dim3(grid), dim3(256), 0, stream, gid, rows, is_key, Dg.ht_slot, Dg.ht_code,
                         Lg.ht_cap[0], groups, code, cnt, miss);
    }
    if (phase == 2) return hipGetLastError();
  }
  if (lds) hipLaunchKernelGGL(seg_prefix_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, stream, hist, slices, (int)groups, cnt);
  hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(1024), 0, stream, cnt, groups, off, uoff, cur);
  const int RS = groups_seg_record_floats(n);
  const unsigned ggrid = (unsigned)cus * 8;
#define SEG_CASE(R, GK)                                                                                               \
  case R: {                                                                                                           \
    constexpr int TW = R > 16 ? 512 : SEG_WG;        /* staged records + 12288 cursors within 160 KB of LDS */      \
    const size_t stage = (size_t)(TW / 64) * (64 * (R + 4) + 64) * 4;                                                 \
    if (lds) {                                                                                                        \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(seg_scatter_kernel<R, true, TW>),             \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                     \
      if (e != hipSuccess) return e;                                                                                  \
      hipLaunchKernelGGL((seg_scatter_kernel<R, true, TW>), dim3(slices), dim3(TW), stage + lds_bytes, stream, code, num, n, rows, \
                         off, hist, (int)groups, cur, rec);                                                           \
    } else {                                                                                                          \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(seg_scatter_kernel<R, false, TW>),            \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                     \
      if (e != hipSuccess) return e;                                                                                  \
      hipLaunchKernelGGL((seg_scatter_kernel<R, false, TW>), dim3((unsigned)std::min<uint64_t>((rows + TW - 1) / TW, (uint64_t)cus)), \
                         dim3(TW), stage, stream, code, num, n, rows, off, hist, (int)groups, cur, rec);              \
    }                                                                                                                 \
    hipLaunchKernelGGL(GK<R>, dim3(ggrid), dim3(256), 0, stream, rec, off, uoff, groups, n, tab, dtot);               \
    break;                                                                                                            \
  }
  switch (RS) {
    SEG_CASE(4, seg_gram_kernel)
    SEG_CASE(8, seg_gram_kernel)
    SEG_CASE(16, seg_gram_kernel)
    SEG_CASE(32, seg_gram_kernel)
    default: return hipErrorInvalidValue;
  }
#undef SEG_CASE
  return hipGetLastError();
}

}  // namespace cofactor
