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

// (experiment knob: COFACTOR_GRAM_CHUNKED)
static int g_gram_chunked = [] { const char *v = getenv("COFACTOR_GRAM_CHUNKED"); return v ? atoi(v) : 0; }();

// T16 (13 <= N <= 16): the whole 16 x 16 matrix of a row as ONE v_mfma_f32_16x16x4_f32 per four rows
// — lane (column i = lane & 15, row kq = lane >> 4 of the group) feeds the same register as A and
// B — instead of ten 4 x 4 block pairs on v_mfma_f32_4x4x1: one ds_read_b32 + one MFMA + one add
// per four rows where the block scheme needs two ds_read_b128 + four MFMAs + two packed adds.
// At 13..16 columns the block scheme sits on its per-row instruction floor below the HBM rate;
// both triangles come out (the lower one is dropped in the fold).
template <int N, bool ALIGNED, bool MASKED, bool T16 = false>
__global__ __launch_bounds__(GRAM_THREADS) void gram_kernel(NumCols cols, uint64_t rows,
                                                            double *__restrict__ partials,
                                                            const uint8_t *__restrict__ mask_arg, int chunked) {
  const uint8_t *__restrict__ mask = MASKED ? mask_arg : nullptr;   // unfiltered variant: no filter bytes are read
  constexpr int NB = (N + 3) / 4;
  constexpr int NPAIR = NB * (NB + 1) / 2;
  constexpr int NBC = T16 ? 16 : 4 * NB;          // data columns incl. zero padding to 4*NB (T16: to 16)
  constexpr int CS = GRAM_COL_STRIDE;
  constexpr int LD = (N * 64 + GRAM_THREADS - 1) / GRAM_THREADS;  // float4 loads / thread / tile
  constexpr int RPM = gram_rows_per_mfma(N);
  constexpr int DEPTH = gram_ring_depth(N);
  // columns [0,N) data, [N,NBC) zero padding, column NBC all-zero (operand of unused blocks)
  constexpr int SCRATCH_FLOATS = (T16 ? 10 : 8) * GRAM_ACC_LEN;   // wave fold: 4 images of doubles (T16: + the result)
  constexpr int TILE_FLOATS = (NBC + 1) * CS > SCRATCH_FLOATS ? (NBC + 1) * CS : SCRATCH_FLOATS;
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
  // T16: this lane's operand of row group g is tile[column lane & 15][wave's row 4 g + (lane >> 4)]
  const float *p16 = tile + (lane & 15) * CS + wave * 64 + (lane >> 4);
  float ls16 = 0.f;
  int since_flush = 0;
  auto crunch = [&]() {
    if constexpr (T16) {
      static_assert(!T16 || (N > 12 && N <= 16), "one 16 x 16 tile");
#pragma unroll
      for (int g = 0; g < 16; g += 4) {           // four independent chains
        const float x0 = p16[4 * g], x1 = p16[4 * g + 4], x2 = p16[4 * g + 8], x3 = p16[4 * g + 12];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, x1, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x2, x2, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x3, x3, acc3, 0, 0, 0);
        ls16 += (x0 + x1) + (x2 + x3);
      }
      // every tile: 4 chains x 4 MFMAs x 4 rows = 64 fp32 adds per cell
      dq0 += (double)((acc0[0] + acc1[0]) + (acc2[0] + acc3[0]));
      dq1 += (double)((acc0[1] + acc1[1]) + (acc2[1] + acc3[1]));
      dq2 += (double)((acc0[2] + acc1[2]) + (acc2[2] + acc3[2]));
      dq3 += (double)((acc0[3] + acc1[3]) + (acc2[3] + acc3[3]));
      dl += (double)ls16;
      acc0 = acc1 = acc2 = acc3 = f32x4{0, 0, 0, 0};
      ls16 = 0.f;
      return;
    }
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
  // tile order: grid-stride (tile b, b + G, ...) or one contiguous run of tiles per workgroup
  uint64_t t = blockIdx.x, tstep = G, tend = nfull;
  if (chunked & 1) {
    const uint64_t per = (nfull + G - 1) / G;
    t = min((uint64_t)blockIdx.x * per, nfull);
    tend = min(t + per, nfull);
    tstep = 1;
  }
  if (t < tend) {
    float4 pre[DEPTH][LD];
    unsigned pmask[DEPTH];
#pragma unroll
    for (int r = 0; r < DEPTH; r++) fetch(pre[r], pmask[r], min(t + r * tstep, tend - 1));
    while (t < tend) {
#pragma unroll
      for (int r = 0; r < DEPTH; r++) {           // tile t sits in ring slot r
#ifdef COFACTOR_DEV_ABLATE
        // 1: no MFMA loop; 2: no park either (barriers stay); 3: bare loads (no park, no barriers)
        if (ablate >= 2) {
          float sink = 0.f;
#pragma unroll
          for (int i = 0; i < LD; i++) sink += pre[r][i].x + pre[r][i].y + pre[r][i].z + pre[r][i].w;
          if (sink == 12345.678f) dl += 1.0;
          if (ablate == 2) __syncthreads();
          fetch(pre[r], pmask[r], min(t + DEPTH * tstep, tend - 1));
          if (ablate == 2) __syncthreads();
          t += tstep;
          if (t >= tend) break;
          continue;
        }
#endif
        park(pre[r], pmask[r]);
        __syncthreads();
        fetch(pre[r], pmask[r], min(t + DEPTH * tstep, tend - 1));   // flies under DEPTH tiles of MFMAs
#ifdef COFACTOR_DEV_ABLATE
        if (ablate == 0)
#endif
        crunch();
        __syncthreads();                          // everyone done reading before the next park()
        t += tstep;
        if (t >= tend) break;
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
  if constexpr (T16) {
    // register r of lane l holds cell (row 4 (l >> 4) + r, column l & 15); lin[c] is spread over the
    // four lanes c, 16 + c, 32 + c, 48 + c.  Into the image layout of device.hpp (fixed order).
    auto over_waves = [&](int j) { return ((red[j] + red[GRAM_ACC_LEN + j]) + red[2 * GRAM_ACC_LEN + j]) + red[3 * GRAM_ACC_LEN + j]; };
    static_assert(!T16 || sizeof(double) * 5 * GRAM_ACC_LEN <= sizeof(float) * TILE_FLOATS, "image scratch must fit the tile");
    double *img = red + 4 * GRAM_ACC_LEN;
    for (int i = tid; i < GRAM_ACC_LEN; i += GRAM_THREADS) img[i] = 0.0;
    __syncthreads();
    for (int e = tid; e < 16 * 16 + 16; e += GRAM_THREADS) {
      if (e < 256) {
        const int j = e >> 4, k = e & 15;
        if (j <= k && k < N) img[gram_quad_pos(j, k, N)] = over_waves((j & 3) * 64 + 16 * (j >> 2) + k);
      } else {
        const int c = e - 256;
        if (c < N)
          img[gram_lin_pos(c, N)] = (over_waves(4 * 64 + c) + over_waves(4 * 64 + 16 + c)) +
                                    (over_waves(4 * 64 + 32 + c) + over_waves(4 * 64 + 48 + c));
      }
    }
    __syncthreads();
    for (int i = tid; i < GRAM_ACC_LEN; i += GRAM_THREADS) partials[(uint64_t)i * gridDim.x + blockIdx.x] = img[i];
    return;
  }
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

// ---- gram_dma_kernel: the same arithmetic, tiles fetched by LDS-DMA (experiment, opt-in) ---------------
// gram_kernel's fetch goes global -> VGPRs -> ds_write ("park") with two barriers per tile; a
// loads-only run of that structure stops at ~5.8 TB/s where the bare loads reach 6.6
// (tests/tools/cols_probe.hip).  Here `global_load_lds_dwordx4` writes each wave's 1 KiB of a
// column straight into a ring of raw tiles in LDS: no park, no VGPR ring, ONE barrier per tile.
// Whole tiles of 16-byte aligned columns without a row filter; launch_gram sends everything else
// (tail, filter, unaligned columns) through gram_kernel.
// Result (20_0, 1e9 rows): 5.0 / 5.7 / 5.95 TB/s at 1 / 2 / 3 workgroups per CU whatever the ring
// depth (2..7) and whether 4 or 8 waves share a tile — the same ~6 TB/s as gram_kernel, so the
// park and the second barrier are not what separates the kernel from the bare loads.  Kept behind
// COFACTOR_GRAM_DMA=1 (tests/test_gpu_fullsize.py runs it).
typedef __attribute__((address_space(3))) void lds_void_t;
constexpr int DMA_COLB = 1040;                   // bytes of one column in a ring slot: 1 KiB + 16 (bank spread)
constexpr int GRAM_RING_MIN_N = 99;              // gram_ring_kernel would be the default from this many columns on (nowhere: see its header)

__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst) {
  // (as an asm statement: the builtin makes hipcc wait vmcnt(0) before the next LDS access, i.e.
  // drain the whole ring; M0 is written in the statement that reads it)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int K>
__device__ __forceinline__ void dma_wait_imm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); }
__device__ __forceinline__ void dma_wait(int k) {          // wave-uniform k
  switch (k) {
#define W(K) case K: dma_wait_imm<K>(); break;
    W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15) W(16) W(17) W(18) W(19)
    W(20) W(21) W(22) W(23) W(24) W(25) W(26) W(27) W(28) W(29) W(30) W(31) W(32) W(33) W(34) W(35) W(36) W(37)
    W(38) W(39) W(40)
#undef W
    default: dma_wait_imm<0>(); break;
  }
}

// WAVES waves share a tile: 256 / WAVES rows each (8 waves: twice the waves per CU for the same LDS,
// half the latency chain per wave and tile).
template <int N, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void gram_dma_kernel(NumCols cols, uint64_t nfull, double *__restrict__ partials,
                                                              int ring) {
  constexpr int THREADS = 64 * WAVES, RW = GRAM_TILE_ROWS / WAVES;   // rows of a tile per wave
  constexpr int NB = (N + 3) / 4;
  constexpr int NPAIR = NB * (NB + 1) / 2;
  constexpr int RPM = gram_rows_per_mfma(N);
  constexpr int SLOT = N * DMA_COLB;
  constexpr int MAXCPW = (N + WAVES - 1) / WAVES;
  static_assert(RW >= 4 * RPM, "a wave needs at least one full round of row groups");
  extern __shared__ __attribute__((aligned(16))) unsigned char dlds[];   // [ring][SLOT] | zero column
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char *zero = dlds + ring * SLOT;
  for (int i = tid; i < DMA_COLB / 4; i += THREADS) reinterpret_cast<unsigned *>(zero)[i] = 0u;

  int colA = -1, colB = -1, rsub = 0;
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
  const bool okA = colA >= 0 && colA < N, okB = colB >= 0 && colB < N;   // else: the zero column
  const int g_row = (wave * RW + 4 * rsub) * 4;
  const int offA = colA * DMA_COLB + g_row, offB = colB * DMA_COLB + g_row;

  // this wave's columns: wave, wave + WAVES, ... (< N); its waits count its own DMA instructions
  const int cpw = wave < N ? (N - wave + WAVES - 1) / WAVES : 0;
  const unsigned lds0 = (unsigned)(unsigned long long)(lds_void_t *)dlds;
  auto dma_tile = [&](uint64_t t, int slot) {
    const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + slot * SLOT);
#pragma unroll
    for (int i = 0; i < MAXCPW; i++)
      if (i < cpw) {
        const int c = wave + WAVES * i;
        dma16(reinterpret_cast<const unsigned char *>(cols.p[c]) + t * (GRAM_TILE_ROWS * 4) + 16 * lane,
              __builtin_amdgcn_readfirstlane(base + c * DMA_COLB));
      }
  };

  f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  f32x2 ls_lo = {0.f, 0.f}, ls_hi = {0.f, 0.f};
  double dq0 = 0, dq1 = 0, dq2 = 0, dq3 = 0, dl = 0;
  int since_flush = 0;
  auto flush = [&]() {
    dq0 += (double)((acc0[0] + acc1[0]) + (acc2[0] + acc3[0]));
    dq1 += (double)((acc0[1] + acc1[1]) + (acc2[1] + acc3[1]));
    dq2 += (double)((acc0[2] + acc1[2]) + (acc2[2] + acc3[2]));
    dq3 += (double)((acc0[3] + acc1[3]) + (acc2[3] + acc3[3]));
    dl += (double)((ls_lo[0] + ls_lo[1]) + (ls_hi[0] + ls_hi[1]));
    acc0 = acc1 = acc2 = acc3 = f32x4{0, 0, 0, 0};
    ls_lo = ls_hi = f32x2{0.f, 0.f};
  };
  __syncthreads();                                           // zero column written

  const uint64_t G = gridDim.x;
  {
    uint64_t t = blockIdx.x;                                 // (blockIdx.x < nfull: the launcher sizes the grid)
    for (int r = 0; r < ring - 1; r++) dma_tile(min(t + r * G, nfull - 1), r);
    int slot = 0;
    const int keep = (ring - 2) * cpw;                       // DMA instructions that may stay in flight at the wait
    for (; t < nfull; t += G) {
      dma_wait(keep);                                        // this wave's part of tile t has landed ...
      __builtin_amdgcn_s_barrier();                          // ... and so has everybody else's; slot - 1 is free
      int nslot = slot + ring - 1;
      nslot = nslot >= ring ? nslot - ring : nslot;
      dma_tile(min(t + (uint64_t)(ring - 1) * G, nfull - 1), nslot);      // past the end: a harmless re-load
      const unsigned char *base = dlds + slot * SLOT;
      const f32x4 *va = reinterpret_cast<const f32x4 *>(okA ? base + offA : zero + g_row);
      const f32x4 *vb = reinterpret_cast<const f32x4 *>(okB ? base + offB : zero + g_row);
#pragma unroll 4
      for (int it = 0; it < RW / 4 / RPM; it++) {
        const f32x4 a = va[it * RPM], b = vb[it * RPM];
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], b[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], b[1], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], b[2], acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], b[3], acc3, 0, 0, 0);
        ls_lo += __builtin_shufflevector(a, a, 0, 1);
        ls_hi += __builtin_shufflevector(a, a, 2, 3);
      }
      if (++since_flush == FLUSH_TILES * RPM) { flush(); since_flush = 0; }
      slot = slot + 1 == ring ? 0 : slot + 1;
    }
  }
  flush();
  dma_wait_imm<0>();                                         // drain the re-loads before the ring is reused
  __syncthreads();

  double *red = reinterpret_cast<double *>(dlds);            // WAVES x 320 doubles (the ring is free)
  double *mine = red + wave * GRAM_ACC_LEN;
  mine[0 * 64 + lane] = dq0;
  mine[1 * 64 + lane] = dq1;
  mine[2 * 64 + lane] = dq2;
  mine[3 * 64 + lane] = dq3;
  mine[4 * 64 + lane] = dl;
  __syncthreads();
  for (int i = tid; i < GRAM_ACC_LEN; i += THREADS) {
    const int ln = i & 63;
    double v = 0;
    if (ln < 4 * NPAIR)
#pragma unroll
      for (int rs = 0; rs < RPM; rs++) {
        const int j = i + 4 * NPAIR * rs;
#pragma unroll
        for (int w = 0; w < WAVES; w++) v += red[w * GRAM_ACC_LEN + j];     // fixed order
      }
    partials[(uint64_t)i * gridDim.x + blockIdx.x] = v;
  }
}

// ---- gram_ring_kernel: LDS-DMA ring with DEDICATED loader waves (default for whole tiles of aligned,
// unfiltered columns) ------------------------------------------------------------------------------------
// gram_dma_kernel's waves both load and compute, so a wave that is late in its MFMAs is late
// with its next DMA and the bytes in flight sag; measured, that kernel needs three workgroups per CU to
// reach gram_kernel's 6.0 TB/s whatever the ring depth.  Here ONE 512-thread workgroup per CU has
//   waves 0-3: loaders — wait (counted vmcnt), barrier, issue the DMAs of the tile `ring - 1` ahead
//              (wave-uniform 64-bit base in SGPRs + one lane-offset VGPR), nothing else;
//   waves 4-7: the Gram of 64 rows each, straight from the raw tile (operand reads issued in batches
//              before their MFMAs: a lone wave hides no LDS latency);
// a loader and a compute wave share each SIMD.  ring - 1 tiles (up to 125 KB at n = 20) stay in
// flight whatever the compute waves do.
// Measured (20_0, 1e9 rows, round 3): the ring ALONE (compute waves idle, COFACTOR_GRAM_RING_ABLATE=1)
// streams 6.75 TB/s = 84 % of the 8 TB/s peak at every depth 3..8; with the compute waves 5.9-6.3 TB/s,
// again at every depth, i.e. the same as gram_kernel (6.0-6.2 on the same box) — the bytes in flight
// are not what limits either kernel; the 128 KB of ds_read_b128 operand reads per 20 KB tile beside
// the DMA writes are the suspect.  Narrow tables lose (n = 4: 5.0 vs 6.2 TB/s: 4 KB tiles, one
// barrier each).  Opt-in: COFACTOR_GRAM_RING=1 (tests/test_gpu_fullsize.py runs it).
__device__ __forceinline__ void dma16_s(const void *sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int N>
__global__ __launch_bounds__(512) void gram_ring_kernel(NumCols cols, uint64_t nfull, double *__restrict__ partials,
                                                        int ring, int ablate) {
  constexpr int NB = (N + 3) / 4;
  constexpr int NPAIR = NB * (NB + 1) / 2;
  constexpr int RPM = gram_rows_per_mfma(N);
  constexpr int SLOT = N * DMA_COLB;
  constexpr int MAXCPW = (N + 3) / 4;
  constexpr int RW = 64;                                     // rows of a tile per compute wave
  static_assert(RW >= 4 * RPM, "a wave needs at least one full round of row groups");
  extern __shared__ __attribute__((aligned(16))) unsigned char dlds[];   // [ring][SLOT] | zero column
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave < 4;
  const int sw = wave & 3;
  unsigned char *zero = dlds + ring * SLOT;
  for (int i = tid; i < DMA_COLB / 4; i += 512) reinterpret_cast<unsigned *>(zero)[i] = 0u;
  __syncthreads();                                           // zero column written
  const uint64_t G = gridDim.x;

  if (loader) {
    // this wave's columns: sw, sw + 4, ... (< N); its waits count its own DMA instructions
    const int cpw = sw < N ? (N - sw + 3) / 4 : 0;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_void_t *)dlds;
    const unsigned lane16 = 16u * lane;
    const unsigned char *src[MAXCPW];
#pragma unroll
    for (int i = 0; i < MAXCPW; i++) src[i] = reinterpret_cast<const unsigned char *>(cols.p[min(sw + 4 * i, N - 1)]);
    auto dma_tile = [&](uint64_t t, int slot) {
      const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + slot * SLOT);
#pragma unroll
      for (int i = 0; i < MAXCPW; i++)
        if (i < cpw) dma16_s(src[i] + t * (GRAM_TILE_ROWS * 4), lane16, __builtin_amdgcn_readfirstlane(base + (sw + 4 * i) * DMA_COLB));
    };
    uint64_t t = blockIdx.x;                                 // (blockIdx.x < nfull: the launcher sizes the grid)
    for (int r = 0; r < ring - 1; r++) dma_tile(min(t + r * G, nfull - 1), r);
    int slot = 0;
    const int keep = (ring - 2) * cpw;                       // DMA instructions that may stay in flight at the wait
    for (; t < nfull; t += G) {
      dma_wait(keep);                                        // this wave's part of tile t has landed ...
      __builtin_amdgcn_s_barrier();                          // ... and so has everybody else's; slot - 1 is free
      int nslot = slot + ring - 1;
      nslot = nslot >= ring ? nslot - ring : nslot;
      dma_tile(min(t + (uint64_t)(ring - 1) * G, nfull - 1), nslot);      // past the end: a harmless re-load
      slot = slot + 1 == ring ? 0 : slot + 1;
    }
    dma_wait_imm<0>();                                       // drain the re-loads before the ring is reused
    __syncthreads();
    __syncthreads();
  } else {
    int colA = -1, colB = -1, rsub = 0;
    {
      const int b = lane >> 2, tt = lane & 3;
      if (b < RPM * NPAIR) {
        rsub = b / NPAIR;
        int bi = 0, rem = b % NPAIR;
        while (rem >= NB - bi) { rem -= NB - bi; bi++; }
        colA = 4 * bi + tt;
        colB = 4 * (bi + rem) + tt;
      }
    }
    const bool okA = colA >= 0 && colA < N, okB = colB >= 0 && colB < N;   // else: the zero column
    const int g_row = (sw * RW + 4 * rsub) * 4;
    const int offA = colA * DMA_COLB + g_row, offB = colB * DMA_COLB + g_row;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    f32x2 ls_lo = {0.f, 0.f}, ls_hi = {0.f, 0.f};
    double dq0 = 0, dq1 = 0, dq2 = 0, dq3 = 0, dl = 0;
    int since_flush = 0;
    auto flush = [&]() {
      dq0 += (double)((acc0[0] + acc1[0]) + (acc2[0] + acc3[0]));
      dq1 += (double)((acc0[1] + acc1[1]) + (acc2[1] + acc3[1]));
      dq2 += (double)((acc0[2] + acc1[2]) + (acc2[2] + acc3[2]));
      dq3 += (double)((acc0[3] + acc1[3]) + (acc2[3] + acc3[3]));
      dl += (double)((ls_lo[0] + ls_lo[1]) + (ls_hi[0] + ls_hi[1]));
      acc0 = acc1 = acc2 = acc3 = f32x4{0, 0, 0, 0};
      ls_lo = ls_hi = f32x2{0.f, 0.f};
    };
    int slot = 0;
    for (uint64_t t = blockIdx.x; t < nfull; t += G) {
      __builtin_amdgcn_s_barrier();                          // tile t is in `slot`
      if (ablate) { slot = slot + 1 == ring ? 0 : slot + 1; continue; }   // (probe: the bare ring, tests/tools)
      const unsigned char *base = dlds + slot * SLOT;
      const f32x4 *va = reinterpret_cast<const f32x4 *>(okA ? base + offA : zero + g_row);
      const f32x4 *vb = reinterpret_cast<const f32x4 *>(okB ? base + offB : zero + g_row);
      // every operand read of the wave's 64 rows is issued up front (<= 128 registers, the wave has 256):
      // one exposed LDS latency per tile, then the MFMAs run as the reads arrive
      constexpr int GIT = RW / 4 / RPM, GBT = GIT;
#pragma unroll
      for (int it0 = 0; it0 < GIT; it0 += GBT) {
        f32x4 ga[GBT], gb[GBT];
#pragma unroll
        for (int it = 0; it < GBT; it++) { ga[it] = va[(it0 + it) * RPM]; gb[it] = vb[(it0 + it) * RPM]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < GBT; it++) {
          const f32x4 a = ga[it], b = gb[it];
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], b[0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], b[1], acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], b[2], acc2, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], b[3], acc3, 0, 0, 0);
          ls_lo += __builtin_shufflevector(a, a, 0, 1);
          ls_hi += __builtin_shufflevector(a, a, 2, 3);
        }
      }
      if (++since_flush == FLUSH_TILES * RPM) { flush(); since_flush = 0; }
      slot = slot + 1 == ring ? 0 : slot + 1;
    }
    flush();
    __syncthreads();                                         // (the loaders' drain: the ring is free)
    double *mine = reinterpret_cast<double *>(dlds) + sw * GRAM_ACC_LEN;
    mine[0 * 64 + lane] = dq0;
    mine[1 * 64 + lane] = dq1;
    mine[2 * 64 + lane] = dq2;
    mine[3 * 64 + lane] = dq3;
    mine[4 * 64 + lane] = dl;
    __syncthreads();
  }
  const double *red = reinterpret_cast<const double *>(dlds);   // 4 x 320 doubles
  for (int i = tid; i < GRAM_ACC_LEN; i += 512) {
    const int ln = i & 63;
    double v = 0;
    if (ln < 4 * NPAIR)
#pragma unroll
      for (int rs = 0; rs < RPM; rs++) {
        const int j = i + 4 * NPAIR * rs;
#pragma unroll
        for (int w = 0; w < 4; w++) v += red[w * GRAM_ACC_LEN + j];     // fixed order
      }
    partials[(uint64_t)i * gridDim.x + blockIdx.x] = v;
  }
}

// ring depth of gram_ring_kernel for n columns: as deep as 150 KB of LDS allow, at most 8 slots and
// at most 40 DMA instructions per loader in flight (the waits are immediates)
static int gram_ring_depth(int n, size_t &lds) {
  static const int env_ring = [] { const char *v = getenv("COFACTOR_GRAM_RING_DEPTH"); return v ? atoi(v) : 0; }();
  const size_t slot = (size_t)n * DMA_COLB;
  const int cpw = (n + 3) / 4;
  int ring = 3;
  while (ring < 8 && (size_t)(ring + 1) * slot + DMA_COLB <= 150 * 1024 && (ring - 1) * cpw <= 40) ring++;
  if (env_ring >= 2 && env_ring <= ring) ring = env_ring;
  lds = std::max((size_t)ring * slot + DMA_COLB, sizeof(double) * 4 * GRAM_ACC_LEN);
  return ring;
}

template <int N>
static hipError_t launch_ring_n(const NumCols &cols, uint64_t nfull, int grid, size_t lds, int ring, double *partials,
                                hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute((const void *)gram_ring_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  static const int ablate = [] { const char *v = getenv("COFACTOR_GRAM_RING_ABLATE"); return v ? atoi(v) : 0; }();
  hipLaunchKernelGGL((gram_ring_kernel<N>), dim3(grid), dim3(512), lds, stream, cols, nfull, partials, ring, ablate);
  return hipGetLastError();
}

// ---- gram_narrow_kernel<N>, N = 1, 2: no tile pipeline, no MFMA ------------------------------------------
// One or two columns are 4-8 bytes per row: gram_kernel's per-tile work (park, barrier, one MFMA per
// row group) bounds it at 2.6 / 4.8 TB/s.  The whole triple of such a table is 2 / 5 numbers, so this
// is the plain streaming reduction in the shape the calibration found fastest (one contiguous range
// per workgroup, 16 workgroups per CU, four 16-byte non-temporal loads in flight per lane), with the
// same arithmetic as everywhere: fp32 products, fp32 chains of <= 64 terms, folded into fp64.
template <int N>
__global__ __launch_bounds__(256) void gram_narrow_kernel(NumCols cols, uint64_t n4, double *__restrict__ partials) {
  constexpr int NQ = N == 1 ? 1 : 3;                        // products: x0 x0 | x0 x0, x0 x1, x1 x1
  const f32x4 *c0 = reinterpret_cast<const f32x4 *>(cols.p[0]);
  const f32x4 *c1 = reinterpret_cast<const f32x4 *>(cols.p[N - 1]);
  const uint64_t per = (n4 / gridDim.x) / 1024 * 1024;
  uint64_t i = (uint64_t)blockIdx.x * per + threadIdx.x;
  const uint64_t end = (uint64_t)blockIdx.x * per + per;
  f32x4 fl[N], fq[NQ];
  double dl[N], dq[NQ];
#pragma unroll
  for (int k = 0; k < N; k++) { fl[k] = f32x4{0, 0, 0, 0}; dl[k] = 0; }
#pragma unroll
  for (int k = 0; k < NQ; k++) { fq[k] = f32x4{0, 0, 0, 0}; dq[k] = 0; }
  int since = 0;
  for (; i < end; i += 1024) {
    f32x4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      a[u] = __builtin_nontemporal_load(c0 + i + 256 * u);
      if (N == 2) b[u] = __builtin_nontemporal_load(c1 + i + 256 * u);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      fl[0] += a[u];
      fq[0] += a[u] * a[u];
      if (N == 2) { fl[N - 1] += b[u]; fq[1] += a[u] * b[u]; fq[NQ - 1] += b[u] * b[u]; }
    }
    if (++since == 16) {                                    // 64 terms per fp32 chain
#pragma unroll
      for (int k = 0; k < N; k++) { dl[k] += (double)((fl[k][0] + fl[k][1]) + (fl[k][2] + fl[k][3])); fl[k] = f32x4{0, 0, 0, 0}; }
#pragma unroll
      for (int k = 0; k < NQ; k++) { dq[k] += (double)((fq[k][0] + fq[k][1]) + (fq[k][2] + fq[k][3])); fq[k] = f32x4{0, 0, 0, 0}; }
      since = 0;
    }
  }
#pragma unroll
  for (int k = 0; k < N; k++) dl[k] += (double)((fl[k][0] + fl[k][1]) + (fl[k][2] + fl[k][3]));
#pragma unroll
  for (int k = 0; k < NQ; k++) dq[k] += (double)((fq[k][0] + fq[k][1]) + (fq[k][2] + fq[k][3]));
  // workgroup sums in a fixed order: lanes by shuffles, waves through LDS
  __shared__ double red[4][N + NQ];
  double v[N + NQ];
#pragma unroll
  for (int k = 0; k < N; k++) v[k] = dl[k];
#pragma unroll
  for (int k = 0; k < NQ; k++) v[N + k] = dq[k];
#pragma unroll
  for (int k = 0; k < N + NQ; k++) {
    for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v[k];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < GRAM_ACC_LEN; idx += 256) partials[(uint64_t)idx * gridDim.x + blockIdx.x] = 0.0;
  __syncthreads();
  if (threadIdx.x < N + NQ) {
    const int k = threadIdx.x;
    const double t = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    int pos;
    if (k < N) pos = gram_lin_pos(k, N);
    else if (N == 1) pos = gram_quad_pos(0, 0, 1);
    else pos = k - N == 0 ? gram_quad_pos(0, 0, 2) : (k - N == 1 ? gram_quad_pos(0, 1, 2) : gram_quad_pos(1, 1, 2));
    partials[(uint64_t)pos * gridDim.x + blockIdx.x] = t;
  }
}

// shape of the DMA kernel for n columns: waves per workgroup, ring depth, workgroups per CU (160 KB
// of LDS per CU)
static void gram_dma_shape(int n, int &waves, int &ring, int &wgs_per_cu, size_t &lds) {
  static const int env_ring = [] { const char *v = getenv("COFACTOR_GRAM_DMA_RING"); return v ? atoi(v) : 0; }();
  static const int env_wgs = [] { const char *v = getenv("COFACTOR_GRAM_DMA_WGS"); return v ? atoi(v) : 0; }();
  static const int env_waves = [] { const char *v = getenv("COFACTOR_GRAM_DMA_WAVES"); return v ? atoi(v) : 0; }();
  waves = (env_waves == 4 || env_waves == 8) ? env_waves : (n > 8 ? 8 : 4);
  if (gram_rows_per_mfma(n) * 4 > GRAM_TILE_ROWS / waves) waves = 4;           // (n <= 4: 64 rows per MFMA round)
  const size_t slot = (size_t)n * DMA_COLB;
  ring = env_ring >= 2 ? env_ring : 2;
  if (!env_ring)
    while (ring < 8 && (size_t)(ring + 1) * slot + DMA_COLB <= 38 * 1024) ring++;   // narrow tables: deeper
  lds = std::max((size_t)ring * slot + DMA_COLB, sizeof(double) * waves * GRAM_ACC_LEN);
  wgs_per_cu = env_wgs > 0 ? env_wgs : (int)std::min<size_t>(4, (160 * 1024) / lds);
  if (wgs_per_cu < 1) wgs_per_cu = 1;
}

template <int N>
static hipError_t launch_dma_n(const NumCols &cols, uint64_t nfull, int grid, int waves, size_t lds, int ring,
                               double *partials, hipStream_t stream) {
  hipError_t e;
  if (waves == 8) {
    if constexpr (gram_rows_per_mfma(N) * 4 <= GRAM_TILE_ROWS / 8) {
      e = hipFuncSetAttribute((const void *)gram_dma_kernel<N, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL((gram_dma_kernel<N, 8>), dim3(grid), dim3(512), lds, stream, cols, nfull, partials, ring);
      return hipGetLastError();
    }
    return hipErrorInvalidValue;
  }
  e = hipFuncSetAttribute((const void *)gram_dma_kernel<N, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((gram_dma_kernel<N, 4>), dim3(grid), dim3(256), lds, stream, cols, nfull, partials, ring);
  return hipGetLastError();
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
  // (up to 12 columns the block scheme packs 2..16 rows into one MFMA and wins: measured 5..12)
  constexpr bool T16 = N > 12 && N <= 16;
  static const bool no16 = [] { const char *v = getenv("COFACTOR_GRAM_NO16"); return v && *v == '1'; }();
  if (T16 && !no16) {
    if constexpr (T16) {
      if (aligned && mask)
        hipLaunchKernelGGL((gram_kernel<N, true, true, true>), dim3(grid), dim3(GRAM_THREADS), 0, stream, cols, rows, partials, mask, g_gram_chunked);
      else if (aligned)
        hipLaunchKernelGGL((gram_kernel<N, true, false, true>), dim3(grid), dim3(GRAM_THREADS), 0, stream, cols, rows, partials, mask, g_gram_chunked);
      else if (mask)
        hipLaunchKernelGGL((gram_kernel<N, false, true, true>), dim3(grid), dim3(GRAM_THREADS), 0, stream, cols, rows, partials, mask, g_gram_chunked);
      else
        hipLaunchKernelGGL((gram_kernel<N, false, false, true>), dim3(grid), dim3(GRAM_THREADS), 0, stream, cols, rows, partials, mask, g_gram_chunked);
    }
    return hipGetLastError();
  }
  if (aligned && mask)
    hipLaunchKernelGGL((gram_kernel<N, true, true>), dim3(grid), dim3(GRAM_THREADS), 0, stream, cols, rows, partials, mask, g_gram_chunked);
  else if (aligned)
    hipLaunchKernelGGL((gram_kernel<N, true, false>), dim3(grid), dim3(GRAM_THREADS), 0, stream, cols, rows, partials, mask, g_gram_chunked);
  else if (mask)
    hipLaunchKernelGGL((gram_kernel<N, false, true>), dim3(grid), dim3(GRAM_THREADS), 0, stream, cols, rows, partials, mask, g_gram_chunked);
  else
    hipLaunchKernelGGL((gram_kernel<N, false, false>), dim3(grid), dim3(GRAM_THREADS), 0, stream, cols, rows, partials, mask, g_gram_chunked);
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
  uint64_t rrows = rows - head;
  if (rrows == 0) return ev1 ? hipEventRecord(ev1, stream) : hipSuccess;
  // COFACTOR_GRAM_DMA=1: whole tiles of aligned, unfiltered columns through the LDS-DMA variant
  // (measured equal to gram_kernel, 13.4 vs 13.5 ms per 1e9 rows at n = 20: off by default)
  // gram_ring_kernel (dedicated loader waves): COFACTOR_GRAM_RING=0 switches it off, =1 forces it for
  // every n; by default it takes the column counts it was measured faster for (g_ring_min_n and up)
  // one or two columns: the plain streaming reduction (gram_narrow_kernel); COFACTOR_GRAM_NARROW=0 switches it off
  static const int narrow_env = [] { const char *v = getenv("COFACTOR_GRAM_NARROW"); return v ? atoi(v) : 1; }();
  if (narrow_env && n <= 2 && !mask && rrows >= ((uint64_t)1 << 22)) {
    bool aligned = true;
    for (int k = 0; k < n; k++) aligned = aligned && ((reinterpret_cast<uintptr_t>(rest.p[k]) & 15) == 0);
    if (aligned) {
      static const int cus = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
      const int ngrid = 16 * cus;                           // (the context's partials hold max(gram grid, 16 x CUs) images)
      const uint64_t n4 = rrows / 4, per = (n4 / ngrid) / 1024 * 1024;
      const uint64_t done = per * (uint64_t)ngrid * 4;
      if (per > 0) {
        if (n == 1) hipLaunchKernelGGL(gram_narrow_kernel<1>, dim3(ngrid), dim3(256), 0, stream, rest, n4, partials);
        else hipLaunchKernelGGL(gram_narrow_kernel<2>, dim3(ngrid), dim3(256), 0, stream, rest, n4, partials);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if (done == rrows) {
          if (ev1 && (e = hipEventRecord(ev1, stream)) != hipSuccess) return e;
          return launch_gram_fold(partials, ngrid, acc, stream);
        }
        if ((e = launch_gram_fold(partials, ngrid, acc, stream)) != hipSuccess) return e;
        for (int k = 0; k < n; k++) rest.p[k] += done;
        rrows -= done;
      }
    }
  }
  static const int ring_env = [] { const char *v = getenv("COFACTOR_GRAM_RING"); return v ? atoi(v) : -1; }();
  static const int dma_env_v = [] { const char *v = getenv("COFACTOR_GRAM_DMA"); return v ? atoi(v) : 0; }();
  const bool want_ring = ring_env == 1 || (ring_env != 0 && !dma_env_v && n >= GRAM_RING_MIN_N);
  if (want_ring && !mask && rrows >= 1024 * (uint64_t)GRAM_TILE_ROWS) {
    bool aligned = true;
    for (int k = 0; k < n; k++) aligned = aligned && ((reinterpret_cast<uintptr_t>(rest.p[k]) & 15) == 0);
    if (aligned) {
      const uint64_t nfull = rrows / GRAM_TILE_ROWS;
      size_t lds;
      const int ring = gram_ring_depth(n, lds);
      static const int cus = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
      int dgrid = std::min(grid, cus);
      if ((uint64_t)dgrid > nfull) dgrid = (int)nfull;
      e = hipErrorInvalidValue;
      switch (n) {
#define CASE(N) case N: e = launch_ring_n<N>(rest, nfull, dgrid, lds, ring, partials, stream); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10)
        CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18) CASE(19) CASE(20)
#undef CASE
        default: break;
      }
      if (e != hipSuccess) return e;
      const uint64_t done = nfull * GRAM_TILE_ROWS;
      if (done == rrows) {
        if (ev1 && (e = hipEventRecord(ev1, stream)) != hipSuccess) return e;
        return launch_gram_fold(partials, dgrid, acc, stream);
      }
      if ((e = launch_gram_fold(partials, dgrid, acc, stream)) != hipSuccess) return e;
      for (int k = 0; k < n; k++) rest.p[k] += done;
      rrows -= done;
    }
  }
  const int use_dma = dma_env_v;
  if (use_dma && !mask && rrows >= 64 * (uint64_t)GRAM_TILE_ROWS) {
    bool aligned = true;
    for (int k = 0; k < n; k++) aligned = aligned && ((reinterpret_cast<uintptr_t>(rest.p[k]) & 15) == 0);
    if (aligned) {
      const uint64_t nfull = rrows / GRAM_TILE_ROWS;
      int waves, ring, wgs;
      size_t lds;
      gram_dma_shape(n, waves, ring, wgs, lds);
      static const int cus = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
      int dgrid = std::min(grid, wgs * cus);
      if ((uint64_t)dgrid > nfull) dgrid = (int)nfull;
      e = hipErrorInvalidValue;
      switch (n) {
#define CASE(N) case N: e = launch_dma_n<N>(rest, nfull, dgrid, waves, lds, ring, partials, stream); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10)
        CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18) CASE(19) CASE(20)
#undef CASE
        default: break;
      }
      if (e != hipSuccess) return e;
      const uint64_t done = nfull * GRAM_TILE_ROWS;
      if (done == rrows) {
        if (ev1 && (e = hipEventRecord(ev1, stream)) != hipSuccess) return e;
        return launch_gram_fold(partials, dgrid, acc, stream);
      }
      if ((e = launch_gram_fold(partials, dgrid, acc, stream)) != hipSuccess) return e;
      for (int k = 0; k < n; k++) rest.p[k] += done;
      rrows -= done;
    }
  }
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

// dst image += src image, dst kept-row counter += src's (SumStateCombine's dense half on the device)
__global__ __launch_bounds__(GRAM_ACC_LEN) void acc_add_kernel(const double *__restrict__ src_acc,
                                                               const unsigned long long *__restrict__ src_kept,
                                                               double *__restrict__ acc, unsigned long long *__restrict__ kept) {
  acc[threadIdx.x] += src_acc[threadIdx.x];
  if (threadIdx.x == 0) *kept += *src_kept;
}
hipError_t launch_acc_add(const double *src_acc, const unsigned long long *src_kept, double *acc,
                          unsigned long long *kept, hipStream_t stream) {
  hipLaunchKernelGGL(acc_add_kernel, dim3(1), dim3(GRAM_ACC_LEN), 0, stream, src_acc, src_kept, acc, kept);
  return hipGetLastError();
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
// float4 per lane, non-temporal, four loads in flight per lane, every workgroup walking its own
// contiguous range, 16 workgroups per CU: the fastest of the shapes tests/tools/read_probe.hip
// sweeps (read-only: 7.0 TB/s; the grid-stride shape at 4 workgroups per CU that this used to be:
// 5.7 TB/s).
__global__ __launch_bounds__(256) void calib_copy_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst,
                                                         uint64_t n4) {
  const uint64_t per = (n4 / gridDim.x) / 1024 * 1024;
  uint64_t i = (uint64_t)blockIdx.x * per + threadIdx.x;
  const uint64_t end = (uint64_t)blockIdx.x * per + per;
  for (; i < end; i += 1024) {
    const f32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + 256);
    const f32x4 c = __builtin_nontemporal_load(src + i + 512), d = __builtin_nontemporal_load(src + i + 768);
    __builtin_nontemporal_store(a, dst + i);
    __builtin_nontemporal_store(b, dst + i + 256);
    __builtin_nontemporal_store(c, dst + i + 512);
    __builtin_nontemporal_store(d, dst + i + 768);
  }
}
__global__ __launch_bounds__(256) void calib_read_kernel(const f32x4 *__restrict__ src, float *__restrict__ out,
                                                         uint64_t n4) {
  const uint64_t per = (n4 / gridDim.x) / 1024 * 1024;
  uint64_t i = (uint64_t)blockIdx.x * per + threadIdx.x;
  const uint64_t end = (uint64_t)blockIdx.x * per + per;
  f32x4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
  for (; i < end; i += 1024) {
    s0 += __builtin_nontemporal_load(src + i);
    s1 += __builtin_nontemporal_load(src + i + 256);
    s2 += __builtin_nontemporal_load(src + i + 512);
    s3 += __builtin_nontemporal_load(src + i + 768);
  }
  const f32x4 s = (s0 + s1) + (s2 + s3);
  const float v = (s[0] + s[1]) + (s[2] + s[3]);
  if (v == 12345.678f) out[0] = v;                 // keeps the loads alive; practically never true
}

// bytes actually moved by one launch (whole 16-KiB steps of every workgroup's range)
uint64_t calibration_bytes(uint64_t bytes, int grid) { return ((bytes / 16) / grid) / 1024 * 1024 * (uint64_t)grid * 16; }

hipError_t launch_calibration(const void *src, void *dst, uint64_t bytes, int grid, bool copy, hipStream_t stream) {
  const uint64_t n4 = bytes / 16;
  if (copy)
    hipLaunchKernelGGL(calib_copy_kernel, dim3(grid), dim3(256), 0, stream, (const f32x4 *)src, (f32x4 *)dst, n4);
  else
    hipLaunchKernelGGL(calib_read_kernel, dim3(grid), dim3(256), 0, stream, (const f32x4 *)src, (float *)dst, n4);
  return hipGetLastError();
}

}  // namespace cofactor
