// sum_to_triple_n_m / sum_to_nb_agg_n_m with low-cardinality key columns (every column <= 16
// distinct keys): ONE pass over the n float and m int32 columns produces the dense part (lin_agg,
// quad_agg) and all categorical tables (lin_cat, quad_num_cat, quad_cat).
//
// Reference loops replaced: duckdb_extension/src/triple/sum/sum_no_lift.cpp:119-214,
// sum_to_nb_agg.cpp:107-145.
//
// Shape of the kernel (DESIGN.md "fused2_kernel"):
//   * one 256-thread workgroup per CU = 4 waves, one per SIMD, each with the whole register file;
//     all waves run the same program (no producer / consumer teams to keep in step);
//   * 256-row tiles arrive by LDS-DMA (global_load_lds_dwordx4, one wave-instruction = 1 KiB = 256
//     rows of ONE column, non-temporal) into a ring of R raw tiles in LDS: no staging registers, so
//     R - 1 tiles (60-100 KB per CU) are in flight whatever the register pressure; one raw
//     s_barrier per tile, the waits are counted (vmcnt((R-2) x loads per tile));
//   * each wave owns 64 rows of the tile.  Prep: keys -> 5-bit codes through a byte table in LDS
//     (hash probe only for keys outside 0..255), one byte per row; floats -> three exact bf16 pieces
//     (x = hi + mid + lo); both into the wave's private scratch, no barrier;
//   * everything categorical is a matrix product of ONE-HOT operands: a lane's 16 code bytes
//     become an int8 one-hot operand with two VALU ops per dword ((c ^ i') + 0x21.. & 0x40..);
//       - pair counts (quad_cat): onehot(c1)^T onehot(c2) on v_mfma_i32_32x32x32_i8, two key columns
//         per operand, so 15 tiles cover the 45 column pairs of m = 10; exact int32 counts;
//       - per-key sums (quad_num_cat) and key counts (lin_cat): onehot^T [pieces | 1] on
//         v_mfma_f32_32x32x16_bf16, the bf16 one-hot being a byte shuffle (v_perm) of the int8 one;
//   * the dense Gram runs on v_mfma_f32_4x4x1 straight from the raw tile, as in gram.hip;
//   * fp32 chains are folded into fp64 (LDS table for the per-key sums, registers for the Gram)
//     before they can lose bits; pair counts leave through a per-workgroup slab and
//     fused_pairs_fold2_kernel.
#include <cstdio>
#include <cstdlib>

#include "onepass.hpp"

namespace cofactor {
using namespace onepass;

namespace {

template <int NBLK, int NBB, int M, int MODE>
__global__ __launch_bounds__(F2_THREADS, 1) void fused2_kernel(NumCols num, CatCols cat, uint64_t rows, CatLayout L,
                                                               CatDevice D, F2Carve cv, int ring,
                                                               double *__restrict__ partials,
                                                               unsigned *__restrict__ pair_slabs,
                                                               unsigned *__restrict__ skip,
                                                               const uint8_t *__restrict__ mask,
                                                               unsigned long long *__restrict__ kept, F2Sub sub) {
  constexpr bool PAIRS = (MODE & F2_PAIRS) != 0, SSUM = (MODE & F2_SSUM) != 0, SUB = (MODE & F2_SUB) != 0;
  constexpr bool GRAM = NBLK > 0 && !SUB;
  constexpr int NPAIR = NBLK * (NBLK + 1) / 2;
  constexpr int NT = PAIRS ? M * (M - 1) / 2 : 0;            // 16x16 pair blocks (c1 < c2)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = L.n, m = L.m;                                // m <= M (M is m rounded up to even)
  const bool masked = mask != nullptr;
  const int ndata = n + m;                                   // DMA'd 1-KiB columns per tile
  const int ncols = ndata + (masked ? 1 : 0);                // + the row filter (256 bytes)
  const int cpw = (ncols + 3) / 4;                           // DMA instructions per wave and tile
  const int pcols = SSUM ? 3 * n : 0;                        // piece columns; column `pcols` is the ones column

  double *l_s = reinterpret_cast<double *>(lds + cv.s);
  unsigned *l_cnt = reinterpret_cast<unsigned *>(lds + cv.cnt);
  unsigned char *l_direct = lds + cv.direct;
  unsigned char *l_far = l_direct + M * DIRECT_STRIDE;
  int *l_hoff = reinterpret_cast<int *>(l_far + 32);         // per column: first dictionary slot, slots (for the probe)
  int *l_hcap = l_hoff + 12;
  unsigned long long *l_slot = reinterpret_cast<unsigned long long *>(lds + cv.slot);
  int32_t *l_dcode = reinterpret_cast<int32_t *>(lds + cv.dcode);
  unsigned char *scratch = lds + cv.scratch + wave * cv.scratch_bytes;   // this wave's codes + pieces
  unsigned char *my_codes = scratch;                         // [M][CST]
  unsigned char *my_pieces = scratch + M * CST;              // [16 * NBB][PST]

  // ---- one-time LDS setup ------------------------------------------------------------------
  for (int i = tid; i < (cv.total - cv.zero) / 4; i += F2_THREADS) reinterpret_cast<unsigned *>(lds + cv.zero)[i] = 0u;
  __syncthreads();
  for (int i = tid; i < L.n_slots; i += F2_THREADS) { l_slot[i] = D.ht_slot[i]; l_dcode[i] = D.ht_code[i]; }
  for (int i = tid; i < M * DIRECT_STRIDE; i += F2_THREADS) l_direct[i] = (unsigned char)NO_CODE;
  if (tid < m) { l_hoff[tid] = L.ht_off[tid]; l_hcap[tid] = L.ht_cap[tid]; }
  {   // code bytes of a column that does not exist (odd m): NO_CODE for ever; ones column: bf16 1.0
    for (int i = lane; i < M * CST; i += 64) my_codes[i] = (unsigned char)NO_CODE;
    unsigned short *ones = reinterpret_cast<unsigned short *>(my_pieces + pcols * PST);
    ones[lane] = 0x3F80;
  }
  __syncthreads();
  for (int c = 0; c < m; c++)
    for (int i = tid; i < L.ht_cap[c]; i += F2_THREADS) {
      const unsigned long long sv = l_slot[L.ht_off[c] + i];
      const int32_t cdv = l_dcode[L.ht_off[c] + i];
      if (sv != 0ull && cdv >= 0) {
        const unsigned key = (unsigned)(sv & 0xFFFFFFFFull);
        if (key < (unsigned)DIRECT_KEYS) l_direct[c * DIRECT_STRIDE + key] = (unsigned char)cdv;
        else l_far[c] = 1;
      }
    }

  // ---- lane roles ----------------------------------------------------------------------------------
  // Gram operand columns as in gram.hip: block b serves block pair b % NPAIR of row group b / NPAIR
  constexpr int RPM = NPAIR <= 1 ? 16 : (NPAIR <= 3 ? 4 : (NPAIR <= 6 ? 2 : 1));
  int colA = -1, colB = -1, rsub = 0;
  if (NBLK > 0) {
    const int b = lane >> 2, t = lane & 3;
    if (b < RPM * NPAIR) {
      rsub = b / NPAIR;
      int bi = 0, rem = b % NPAIR;
      while (rem >= NBLK - bi) { rem -= NBLK - bi; bi++; }
      colA = 4 * bi + t;
      colB = 4 * (bi + rem) + t;
    }
  }
  // byte offsets of this lane's Gram operands inside a slot; a column >= n reads the zero column
  const bool okA = colA >= 0 && colA < n, okB = colB >= 0 && colB < n;
  const int g_row = (wave * 64 + 4 * rsub) * 4;
  const int offA = colA * COLB + g_row, offB = colB * COLB + g_row;
  // one-hot operands (16x16 MFMAs): lane (i = lane & 15, q = lane >> 4) holds code value i for the
  // rows 16 q .. 16 q + 15 of the wave's 64 rows.  (code ^ i ^ 31) is 31 exactly on a match, and
  // adding 0x21 carries into bit 6 exactly then (codes are 0..17: no carry between the bytes).
  const int li = lane & 15, lq = lane >> 4;
  const unsigned ixor = (unsigned)(li ^ 31) * 0x01010101u;

  const uint64_t ntiles = rows / TR;
  const uint64_t G = gridDim.x;

  // ---- tile ring -----------------------------------------------------------------------------------
  const unsigned lds0 = (unsigned)(unsigned long long)(lds_void *)lds;   // LDS byte address of the block
  // this wave's columns: virtual column min(wave + 4 i, ncols - 1) for i < cpw (a slot past the last
  // column re-loads it: every wave issues the same number of loads, the waits are counted)
  constexpr int MAXCPW = (4 * NBLK + M + 1 + 3) / 4;
  const unsigned char *dsrc[MAXCPW];
  unsigned doff[MAXCPW];
#pragma unroll
  for (int i = 0; i < MAXCPW; i++) {
    const int vc = min(wave + 4 * i, ncols - 1);
    dsrc[i] = vc < n ? reinterpret_cast<const unsigned char *>(num.p[min(vc, COFACTOR_MAX_NUM - 1)])
                     : (vc < ndata ? reinterpret_cast<const unsigned char *>(cat.p[min(max(vc - n, 0), COFACTOR_MAX_CAT - 1)])
                                   : reinterpret_cast<const unsigned char *>(mask));
    doff[i] = (unsigned)(vc * COLB);
  }
  auto dma_tile = [&](uint64_t t, int slot) {
    const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + cv.ring + slot * cv.slot_bytes);
#pragma unroll
    for (int i = 0; i < MAXCPW; i++)
      if (i < cpw) {
        if (!masked || wave + 4 * i < ndata)
          glds16(dsrc[i] + t * (TR * 4) + 16 * lane, __builtin_amdgcn_readfirstlane(base + doff[i]));
        else
          glds4(dsrc[i] + t * TR + 4 * lane, __builtin_amdgcn_readfirstlane(base + doff[i]));
      }
  };

  // ---- accumulators ----------------------------------------------------------------------------------
  i32x4 pacc[NT > 0 ? NT : 1];
  f32x4 sacc[M][NBB];
#pragma unroll
  for (int q = 0; q < (NT > 0 ? NT : 1); q++) pacc[q] = i32x4{0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < M; c++)
#pragma unroll
    for (int bb = 0; bb < NBB; bb++) sacc[c][bb] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  f32x2 ls_lo = {0.f, 0.f}, ls_hi = {0.f, 0.f};
  double dq0 = 0, dq1 = 0, dq2 = 0, dq3 = 0, dl = 0;
  unsigned n_kept = 0;

  auto flush_gram = [&]() {
    dq0 += (double)((acc0[0] + acc1[0]) + (acc2[0] + acc3[0]));
    dq1 += (double)((acc0[1] + acc1[1]) + (acc2[1] + acc3[1]));
    dq2 += (double)((acc0[2] + acc1[2]) + (acc2[2] + acc3[2]));
    dq3 += (double)((acc0[3] + acc1[3]) + (acc2[3] + acc3[3]));
    dl += (double)((ls_lo[0] + ls_lo[1]) + (ls_hi[0] + ls_hi[1]));
    acc0 = acc1 = acc2 = acc3 = f32x4{0, 0, 0, 0};
    ls_lo = ls_hi = f32x2{0.f, 0.f};
  };
  // D register r of lane (li, lq) is cell (A-row 4 lq + r, B-column li): key code 4 lq + r of column
  // c, piece column 16 bb + li.  The bf16 one-hot is 2.0 (0x4000), hence the 0.5.
  auto flush_s = [&]() {
#pragma unroll
    for (int c = 0; c < M; c++)
#pragma unroll
      for (int bb = 0; bb < NBB; bb++) mfma_settle(sacc[c][bb]);
#pragma unroll
    for (int bb = 0; bb < NBB; bb++) {
      const int pc = 16 * bb + li;
      const bool is_sum = pc < pcols, is_cnt = pc == pcols;
      const int k = is_sum ? pc % n : 0;
#pragma unroll
      for (int c = 0; c < M; c++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float v = sacc[c][bb][r];
          if (c < m && v != 0.f) {
            const int code = 4 * lq + r;
            if (is_sum) unsafeAtomicAdd(&l_s[L.s_off[c] + code * n + k], (double)v * 0.5);
            else if (is_cnt) atomicAdd(&l_cnt[16 * c + code], (unsigned)(v * 0.5f));
          }
        }
        sacc[c][bb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    asm volatile("" ::: "memory");
  };

  // ---- one wave's 64 rows of the tile in slot `slot` ------------------------------------------------
  auto subtile = [&](uint64_t t, int slot) {
    unsigned char *base = lds + cv.ring + slot * cv.slot_bytes;
    const unsigned char *mrow = base + ndata * COLB + wave * 64;          // this wave's 64 filter bytes
    const int q4 = 4 * (lane & 15);                                       // this lane's 4 rows in the prep passes
    unsigned fl = 0x01010101u;
    if (masked) fl = *reinterpret_cast<const unsigned *>(mrow + q4);
    // -- keys -> codes (one byte per row).  Branch-free and unrolled so that the LDS round trips of
    //    the column groups overlap: lanes past the last column redo it (same bytes, same place) --
    bool unknown = false;
#pragma unroll
    for (int j = 0; j < (M + 3) / 4; j++) {
      const int c = min(4 * j + (lane >> 4), m - 1);
      const uint4 kv = *reinterpret_cast<const uint4 *>(base + (n + c) * COLB + (wave * 64 + q4) * 4);
      const unsigned char *dt = l_direct + c * DIRECT_STRIDE;             // byte table: keys 0..255, [256] = NO_CODE
      unsigned cx = dt[min(kv.x, (unsigned)DIRECT_KEYS)], cy = dt[min(kv.y, (unsigned)DIRECT_KEYS)];
      unsigned cz = dt[min(kv.z, (unsigned)DIRECT_KEYS)], cw = dt[min(kv.w, (unsigned)DIRECT_KEYS)];
      if (l_far[c] && ((cx | cy | cz | cw) & NO_CODE)) {                  // the column holds keys outside 0..255: probe
        const unsigned long long *sl = l_slot + l_hoff[c];
        const int32_t *dc = l_dcode + l_hoff[c];
        const int cap = l_hcap[c];
        if (cx == NO_CODE) cx = lds_lookup1(sl, dc, cap, kv.x);
        if (cy == NO_CODE) cy = lds_lookup1(sl, dc, cap, kv.y);
        if (cz == NO_CODE) cz = lds_lookup1(sl, dc, cap, kv.z);
        if (cw == NO_CODE) cw = lds_lookup1(sl, dc, cap, kv.w);
      }
      // dropped rows match no code and take no part in the check
      cx = (fl & 0x000000FFu) ? cx : ROW_OFF; cy = (fl & 0x0000FF00u) ? cy : ROW_OFF;
      cz = (fl & 0x00FF0000u) ? cz : ROW_OFF; cw = (fl & 0xFF000000u) ? cw : ROW_OFF;
      unknown = unknown || cx == NO_CODE || cy == NO_CODE || cz == NO_CODE || cw == NO_CODE;
      *reinterpret_cast<unsigned *>(my_codes + c * CST + q4) = cx | (cy << 8) | (cz << 16) | (cw << 24);
    }
    if (__builtin_amdgcn_ballot_w64(unknown) != 0ull) {
      // optimistic mode: the 64 rows are left out as a whole and redone by the host after a
      // dictionary pass; otherwise the dictionary pass has missed a key (reported at the next sync).
      // Left out = every row filtered: no code matches, every x is 0 (the products below still run:
      // a branch around them would make the accumulators values of two paths).
      if (lane == 0) {
        if (skip) skip[1 + atomicAdd(&skip[0], 1u)] = (unsigned)(t * 4 + wave);
        else D.flags[1] = 1;
      }
      wait_vmcnt_imm<0>();                                   // (rare path: keep the hand-counted waits exact)
      fl = 0u;
      for (int c0 = 0; c0 < m; c0 += 4) {
        const int c = min(c0 + (lane >> 4), m - 1);
        *reinterpret_cast<unsigned *>(my_codes + c * CST + q4) = ROW_OFF * 0x01010101u;
      }
    }
    const bool some_dropped = __builtin_amdgcn_ballot_w64(fl != 0x01010101u) != 0ull;   // wave-uniform
    if (masked && lane < 16)                                 // lanes 0..15 hold the 64 filter bytes once
      n_kept += ((fl & 0x000000FFu) != 0) + ((fl & 0x0000FF00u) != 0) + ((fl & 0x00FF0000u) != 0) + ((fl & 0xFF000000u) != 0);
    // -- floats -> bf16 pieces; dropped rows become zeros in the raw tile too (the Gram reads it) --
    bool nonfinite = false;
    if (SSUM || some_dropped) {
#pragma unroll
      for (int j = 0; j < NBLK; j++) {
        const int c = min(4 * j + (lane >> 4), n - 1);                    // lanes past the last column redo it
        uint4 *src = reinterpret_cast<uint4 *>(base + c * COLB + (wave * 64 + q4) * 4);
        uint4 xv = *src;
        unsigned u[4] = {xv.x, xv.y, xv.z, xv.w};
        if (some_dropped) {
          u[0] = (fl & 0x000000FFu) ? u[0] : 0u; u[1] = (fl & 0x0000FF00u) ? u[1] : 0u;
          u[2] = (fl & 0x00FF0000u) ? u[2] : 0u; u[3] = (fl & 0xFF000000u) ? u[3] : 0u;
          *src = make_uint4(u[0], u[1], u[2], u[3]);
        }
        if (SSUM) {
          // x = hi + mid + lo, each a bf16 (exact): hi = upper half of x, mid = upper half of
          // x - hi, lo = x - hi - mid (its lower half is zero)
          constexpr unsigned UPPER_HALVES = 0x07060302u;                  // {s0.b3, s0.b2, s1.b3, s1.b2}
          float r1[4], r2[4];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            r1[e] = __uint_as_float(u[e]) - __uint_as_float(u[e] & 0xFFFF0000u);
            r2[e] = r1[e] - __uint_as_float(__float_as_uint(r1[e]) & 0xFFFF0000u);
          }
          uint2 ph = make_uint2(__builtin_amdgcn_perm(u[1], u[0], UPPER_HALVES), __builtin_amdgcn_perm(u[3], u[2], UPPER_HALVES));
          uint2 pm = make_uint2(__builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), UPPER_HALVES),
                                __builtin_amdgcn_perm(__float_as_uint(r1[3]), __float_as_uint(r1[2]), UPPER_HALVES));
          uint2 pl = make_uint2(__builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), UPPER_HALVES),
                                __builtin_amdgcn_perm(__float_as_uint(r2[3]), __float_as_uint(r2[2]), UPPER_HALVES));
          // inf / nan (rare): pieces 0 here, the value is added to its own key's cells below
          // (0 x inf would poison every key's cell).  x - hi is nan exactly for those.
          const unsigned bad = ((u[0] & 0x7F800000u) == 0x7F800000u) | (((u[1] & 0x7F800000u) == 0x7F800000u) << 1) |
                               (((u[2] & 0x7F800000u) == 0x7F800000u) << 2) | (((u[3] & 0x7F800000u) == 0x7F800000u) << 3);
          if (bad) {
            nonfinite = true;
            const unsigned k0 = ((bad & 1) ? 0u : 0x0000FFFFu) | ((bad & 2) ? 0u : 0xFFFF0000u);
            const unsigned k1 = ((bad & 4) ? 0u : 0x0000FFFFu) | ((bad & 8) ? 0u : 0xFFFF0000u);
            ph.x &= k0; pm.x &= k0; pl.x &= k0;
            ph.y &= k1; pm.y &= k1; pl.y &= k1;
          }
          *reinterpret_cast<uint2 *>(my_pieces + c * PST + 2 * q4) = ph;
          *reinterpret_cast<uint2 *>(my_pieces + (n + c) * PST + 2 * q4) = pm;
          *reinterpret_cast<uint2 *>(my_pieces + (2 * n + c) * PST + 2 * q4) = pl;
        }
      }
    }
    // the prep stores of this wave are read by other lanes of the SAME wave below: LDS executes a
    // wave's instructions in order; this only keeps the compiler from moving loads above the stores
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (SSUM && __builtin_amdgcn_ballot_w64(nonfinite) != 0ull) {
      // rare: add every inf / nan of these 64 rows straight to its keys' cells
      for (int k = 0; k < n; k++) {
        const float x = *reinterpret_cast<const float *>(base + k * COLB + (wave * 64 + lane) * 4);
        if ((__float_as_uint(x) & 0x7F800000u) == 0x7F800000u)
          for (int c = 0; c < m; c++) {
            const unsigned cd = my_codes[c * CST + lane];
            if (cd < 16u) unsafeAtomicAdd(&l_s[L.s_off[c] + (int)cd * n + k], (double)x);
          }
      }
    }

    // (scheduling fences between the phases: with the whole register file at its disposal hipcc
    // otherwise hoists every LDS read of the tile to the top and spills the accumulators)
    __builtin_amdgcn_sched_barrier(0);
    // -- the dense Gram straight from the raw tile --
    if (GRAM) {
      const f32x4 *va = reinterpret_cast<const f32x4 *>(okA ? base + offA : lds + cv.zero + g_row);
      const f32x4 *vb = reinterpret_cast<const f32x4 *>(okB ? base + offB : lds + cv.zero + g_row);
#pragma unroll
      for (int it = 0; it < 16 / RPM; it++) {
        const f32x4 a = va[it * RPM], bv = vb[it * RPM];
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], bv[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], bv[1], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], bv[2], acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], bv[3], acc3, 0, 0, 0);
        ls_lo += __builtin_shufflevector(a, a, 0, 1);
        ls_hi += __builtin_shufflevector(a, a, 2, 3);
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // -- one-hot products over the wave's 64 rows.  k index: lane (li, lq), byte j <-> row 16 lq + j (A and
    //    B operands are built the same way, so any k order of the instruction pairs the same rows) --
    i32x4 oh[M];
#pragma unroll
    for (int c = 0; c < M; c++) {
      const uint4 cb = *reinterpret_cast<const uint4 *>(my_codes + c * CST + 16 * lq);
      oh[c][0] = (int)(xad(cb.x, ixor, 0x21212121u) & 0x40404040u);
      oh[c][1] = (int)(xad(cb.y, ixor, 0x21212121u) & 0x40404040u);
      oh[c][2] = (int)(xad(cb.z, ixor, 0x21212121u) & 0x40404040u);
      oh[c][3] = (int)(xad(cb.w, ixor, 0x21212121u) & 0x40404040u);
    }
    // (an asm MFMA must not read a register the VALU instruction right before it wrote: every
    // one-hot register is an operand of this statement, so all of them are complete before it)
    settle_operands<M>(oh);
    if (PAIRS) {
      int q = 0;
#pragma unroll
      for (int c1 = 0; c1 < M; c1++)
#pragma unroll
        for (int c2 = c1 + 1; c2 < M; c2++, q++)
          pmfma(pacc[q], oh[c1], oh[c2]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; s++) {                            // bf16 k = 32: rows 16 lq + 8 s + j, j = 0..7
      u32x4 bop[NBB];
#pragma unroll
      for (int bb = 0; bb < NBB; bb++)
        bop[bb] = *reinterpret_cast<const u32x4 *>(my_pieces + (16 * bb + li) * PST + 2 * (16 * lq + 8 * s));
      // software pipeline: column c + 1's operand is shuffled before column c's MFMAs are issued
      u32x4 ab = onehot_bf16((unsigned)oh[0][2 * s], (unsigned)oh[0][2 * s + 1]);
#pragma unroll
      for (int c = 0; c < M; c++) {
        u32x4 nx = ab;
        if (c + 1 < M) nx = onehot_bf16((unsigned)oh[c + 1][2 * s], (unsigned)oh[c + 1][2 * s + 1]);
#pragma unroll
        for (int bb = 0; bb < NBB; bb++) smfma(sacc[c][bb], ab, bop[bb]);
        ab = nx;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- the pipeline: ring of `ring` raw tiles, one barrier per tile ----------------------------------
  __syncthreads();                                           // LDS setup visible
  {
    uint64_t t = blockIdx.x;                                 // (blockIdx.x < ntiles: the launcher sizes the grid)
    for (int r = 0; r < ring - 1; r++) dma_tile(min(t + r * G, ntiles - 1), r);
    int slot = 0, since_g = 0, since_s = 0;
    const int keep = (ring - 2) * cpw;                       // DMA instructions that may stay in flight at the wait
    for (; t < ntiles; t += G) {
      wait_vmcnt(keep);                                      // this wave's part of tile t has landed ...
      __builtin_amdgcn_s_barrier();                          // ... and so has everybody else's; slot - 1 is free
      int nslot = slot + ring - 1;
      nslot = nslot >= ring ? nslot - ring : nslot;
      dma_tile(min(t + (uint64_t)(ring - 1) * G, ntiles - 1), nslot);     // past the end: a harmless re-load
      subtile(t, slot);
      if (GRAM && ++since_g == G_FLUSH_TILES * RPM) { flush_gram(); since_g = 0; }
      if (++since_s == S_FLUSH_TILES) { flush_s(); since_s = 0; }
      slot = slot + 1 == ring ? 0 : slot + 1;
    }
    if (GRAM) flush_gram();
    flush_s();
  }
  wait_vmcnt_imm<0>();                                       // drain the re-loads before the ring is reused
  __syncthreads();

  // ---- end: pair counts of the 4 waves -> the workgroup's slab; Gram image; count / sum tables -------
  if (PAIRS) {
#pragma unroll
    for (int q = 0; q < NT; q++) mfma_settle(pacc[q]);
    unsigned *l_p = reinterpret_cast<unsigned *>(lds + cv.ring);          // (the ring is free now)
    for (int i = tid; i < L.n_p; i += F2_THREADS) l_p[i] = 0u;
    __syncthreads();
    int q = 0;
#pragma unroll
    for (int c1 = 0; c1 < M; c1++)
#pragma unroll
      for (int c2 = c1 + 1; c2 < M; c2++, q++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const unsigned v = (unsigned)pacc[q][r] >> 12;                  // the one-hots are 64: 64 x 64 per row
          if (c2 < m && v) {
            const int qi = c1 * m - c1 * (c1 - 1) / 2 + (c2 - c1);
            atomicAdd(&l_p[256 * qi + 16 * (4 * lq + r) + li], v);
          }
        }
    __syncthreads();
    unsigned *slab = pair_slabs + (uint64_t)blockIdx.x * (uint64_t)L.n_p;
    for (int i = tid; i < L.n_p; i += F2_THREADS) slab[i] = l_p[i];
  }
  if (GRAM) {
    __syncthreads();
    double *red = reinterpret_cast<double *>(lds + cv.ring) + (PAIRS ? (L.n_p + 1) / 2 : 0);
    double *mine = red + wave * GRAM_ACC_LEN;
    mine[0 * 64 + lane] = dq0; mine[1 * 64 + lane] = dq1; mine[2 * 64 + lane] = dq2; mine[3 * 64 + lane] = dq3;
    mine[4 * 64 + lane] = dl;
    __syncthreads();
    for (int i = tid; i < GRAM_ACC_LEN; i += F2_THREADS) {   // 4 waves and RPM row groups, fixed order
      double v = 0;
      if ((i & 63) < 4 * NPAIR)
#pragma unroll
        for (int rs = 0; rs < RPM; rs++) {
          const int j = i + 4 * NPAIR * rs;
          v += ((red[j] + red[GRAM_ACC_LEN + j]) + red[2 * GRAM_ACC_LEN + j]) + red[3 * GRAM_ACC_LEN + j];
        }
      partials[(uint64_t)i * gridDim.x + blockIdx.x] = v;
    }
  }
  if (SUB) {
    // the group's tables into the aggregate's: counts once per group of key columns, sums at
    // numeric column sub.k0 + k of the full-width rows
    if (sub.do_cnt)
      for (int i = tid; i < 16 * m; i += F2_THREADS)
        if (l_cnt[i]) atomicAdd(&D.cnt[sub.cnt_goff[i >> 4] + (i & 15)], (unsigned long long)l_cnt[i]);
    if (SSUM)
      for (int i = tid; i < L.n_s; i += F2_THREADS)
        if (l_s[i] != 0.0) {
          const int c = i / (16 * n), rem = i - c * 16 * n;
          const int code = rem / n, k = rem - code * n;
          unsafeAtomicAdd(&D.s[sub.s_goff[c] + code * sub.n_full + sub.k0 + k], l_s[i]);
        }
    return;
  }
  // key counts: to cnt and to the diagonal cells (k, k) of the column's own pair table
  for (int i = tid; i < 16 * m; i += F2_THREADS)
    if (l_cnt[i]) {
      atomicAdd(&D.cnt[i], (unsigned long long)l_cnt[i]);
      if (L.kind == 0) {
        const int c = i >> 4, code = i & 15;
        const int qd = c * m - c * (c - 1) / 2;
        atomicAdd(&D.p[L.p_off[qd] + code * 16 + code], (unsigned long long)l_cnt[i]);
      }
    }
  if (SSUM)
    for (int i = tid; i < L.n_s; i += F2_THREADS)
      if (l_s[i] != 0.0) unsafeAtomicAdd(&D.s[i], l_s[i]);
  if (masked) {
    unsigned long long kk = n_kept;
    for (int off = 32; off > 0; off >>= 1) kk += __shfl_down(kk, off, 64);
    if (lane == 0 && kk && kept) atomicAdd(kept, kk);
  }
}

// D.p[cell] += sum over workgroups of slab[wg][cell]
__global__ __launch_bounds__(256) void fused_pairs_fold2_kernel(const unsigned *__restrict__ slabs, int nwg, int n_p,
                                                                unsigned long long *__restrict__ p) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= n_p) return;
  unsigned long long total = 0;
  for (int w = 0; w < nwg; w++) total += slabs[(uint64_t)w * n_p + cell];
  if (total) p[cell] += total;
}

// temp[col][i * unit + j] = col[list[i] * unit + j]: the row blocks the optimistic pass left out, packed
__global__ __launch_bounds__(64) void gather_units_kernel(NumCols num, CatCols cat, int n, int m, int unit,
                                                          const unsigned *__restrict__ list, unsigned count,
                                                          unsigned *__restrict__ temp, uint64_t temp_stride,
                                                          const uint8_t *__restrict__ mask,
                                                          uint8_t *__restrict__ temp_mask) {
  for (unsigned i = blockIdx.x; i < count; i += gridDim.x)
    for (int j = threadIdx.x; j < unit; j += blockDim.x) {
      const uint64_t src = (uint64_t)list[i] * unit + j;
      const uint64_t dst = (uint64_t)i * unit + j;
      if (mask) temp_mask[dst] = mask[src];
      for (int c = 0; c < n + m; c++) {
        const unsigned *col = c < n ? reinterpret_cast<const unsigned *>(num.p[c])
                                    : reinterpret_cast<const unsigned *>(cat.p[c - n]);
        temp[(uint64_t)c * temp_stride + dst] = col[src];
      }
    }
}

struct F2Shape { int nblk, nbb, me, mode; };   // me = m rounded up to even (template parameter M)

// S-accumulator blocks one wave may hold in VGPRs next to its operands (4 registers each)
constexpr int F2_MAX_SBLOCKS = 24;

bool f2_shape(const CatLayout &L, F2Shape &sh) {
  if (L.m < 1 || L.m > 10 || L.n < 0 || L.n > COFACTOR_MAX_NUM) return false;
  sh.nblk = (L.n + 3) / 4;
  sh.me = (L.m + 1) / 2 * 2;
  if (L.kind == 0) {
    sh.mode = F2_PAIRS | (L.n > 0 ? F2_SSUM : 0);
    sh.nbb = (3 * L.n + 1 + 15) / 16;             // piece columns + the ones column, 16 per block
  } else {
    sh.mode = 0;                                  // NB: key counts and the Gram diagonal only
    sh.nbb = 1;
  }
  return sh.me * sh.nbb <= F2_MAX_SBLOCKS;
}

F2Carve f2_carve(const CatLayout &L, const F2Shape &sh, bool masked, int ring) {
  F2Carve c{};
  size_t o = 0;
  auto take = [&](size_t bytes, size_t align) { o = (o + align - 1) / align * align; size_t at = o; o += bytes; return (int)at; };
  c.slot_bytes = (L.n + L.m) * COLB + (masked ? 256 : 0);
  c.slot_bytes = (c.slot_bytes + 15) / 16 * 16;
  const size_t ring_bytes = (size_t)ring * c.slot_bytes;
  // at the end the ring area holds the workgroup's pair table and the 4 Gram images
  const size_t tail_bytes = (size_t)(sh.mode & F2_PAIRS ? (L.n_p + 1) / 2 * 8 : 0) + sizeof(double) * 4 * GRAM_ACC_LEN;
  c.ring = take(ring_bytes > tail_bytes ? ring_bytes : tail_bytes, 16);
  c.zero = take(COLB, 16);                        // (everything from here on is zeroed at start)
  c.scratch_bytes = sh.me * CST + 16 * sh.nbb * PST;
  c.scratch = take((size_t)4 * c.scratch_bytes, 16);
  c.s = take((size_t)(sh.mode & F2_SSUM ? L.n_s : 0) * 8, 8);
  c.cnt = take((size_t)16 * L.m * 4, 4);
  c.direct = take((size_t)sh.me * DIRECT_STRIDE + 32 + 24 * 4, 4);
  c.slot = take((size_t)L.n_slots * 8, 8);
  c.dcode = take((size_t)L.n_slots * 4, 4);
  c.total = (int)((o + 15) / 16 * 16);
  return c;
}

// Workgroups per CU: the pair accumulators fill a wave's whole register file (one workgroup per
// CU); the modes without pair tables need at most 256 registers per wave and run two workgroups
// per CU — one wave per SIMD hides no latency at all.  COFACTOR_F2_WGS overrides (experiments).
int f2_wgs(const F2Shape &sh) {
  if (sh.mode & F2_PAIRS) return 1;
  static const int w = [] { const char *v = getenv("COFACTOR_F2_WGS"); return v ? atoi(v) : 2; }();   // (read once)
  return w < 1 ? 1 : (w > 3 ? 3 : w);
}

// deepest ring that fits `wgs` workgroups into the CU's LDS: at most 8 slots, at least 3 (2 tiles
// in flight) for a lone workgroup, 2 when several share the CU (their loads overlap each other)
int f2_ring_for(const CatLayout &L, const F2Shape &sh, bool masked, size_t lds_limit, int wgs) {
  int best = 0;
  for (int r = wgs > 1 ? 2 : 3; r <= 8; r++) {
    const F2Carve c = f2_carve(L, sh, masked, r);
    const int cpw = (L.n + L.m + (masked ? 1 : 0) + 3) / 4;
    if ((size_t)c.total * wgs <= lds_limit && (r - 2) * cpw <= 48) best = r;
  }
  return best;
}
// ring depth and workgroups per CU of a launch
int f2_ring(const CatLayout &L, const F2Shape &sh, bool masked, size_t lds_limit, int *wgs_out = nullptr) {
  for (int wgs = f2_wgs(sh); wgs >= 1; wgs--) {
    const int r = f2_ring_for(L, sh, masked, lds_limit, wgs);
    if (r >= 2) { if (wgs_out) *wgs_out = wgs; return r; }
  }
  if (wgs_out) *wgs_out = 1;
  return 0;
}

template <int NBLK, int NBB, int M, int MODE>
hipError_t f2_launch_one(const NumCols &num, const CatCols &cat, uint64_t rows, const CatLayout &L,
                         const CatDevice &D, const F2Carve &cv, int ring, int grid, double *partials,
                         unsigned *slabs, unsigned *skip, const uint8_t *mask, unsigned long long *kept,
                         hipStream_t stream, const F2Sub &sub = F2Sub{}) {
  hipError_t e = hipFuncSetAttribute((const void *)fused2_kernel<NBLK, NBB, M, MODE>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, cv.total);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((fused2_kernel<NBLK, NBB, M, MODE>), dim3(grid), dim3(F2_THREADS), cv.total, stream,
                     num, cat, rows, L, D, cv, ring, partials, slabs, skip, mask, kept, sub);
  return hipGetLastError();
}

template <int NBLK, int NBB, int MODE>
hipError_t f2_launch_m(int me, const NumCols &num, const CatCols &cat, uint64_t rows,
                       const CatLayout &L, const CatDevice &D, const F2Carve &cv, int ring, int grid,
                       double *partials, unsigned *slabs, unsigned *skip, const uint8_t *mask,
                       unsigned long long *kept, hipStream_t stream, const F2Sub &sub = F2Sub{}) {
  switch (me) {
#define CASE(M_) case M_: if constexpr (M_ * NBB <= F2_MAX_SBLOCKS) \
      return f2_launch_one<NBLK, NBB, M_, MODE>(num, cat, rows, L, D, cv, ring, grid, partials, slabs, skip, mask, kept, stream, sub); \
    else break;
    CASE(2) CASE(4) CASE(6) CASE(8) CASE(10)
#undef CASE
    default: break;
  }
  return hipErrorInvalidValue;
}

}  // namespace

bool fused2_applicable(const CatLayout &L, const int32_t *nkeys, bool masked, size_t lds_limit) {
  for (int c = 0; c < L.m; c++)
    if (nkeys[c] > 16 || L.kc[c] != 16) return false;
  F2Shape sh;
  if (!f2_shape(L, sh)) return false;
  return f2_ring(L, sh, masked, lds_limit) >= 2;
}

int fused2_grid(int cus, int partials_cap_wgs, uint64_t rows, int wgs_per_cu) {
  int grid = cus * (wgs_per_cu < 1 ? 1 : wgs_per_cu);         // the kernel claims 1 / wgs_per_cu of a CU's LDS
  if (grid > partials_cap_wgs) grid = partials_cap_wgs;
  const uint64_t ntiles = rows / TR;
  if ((uint64_t)grid > ntiles) grid = (int)ntiles;
  return grid;
}

int fused2_wgs_per_cu(const CatLayout &L, bool masked, size_t lds_limit) {
  F2Shape sh;
  int wgs = 1;
  if (!f2_shape(L, sh)) return 1;
  (void)f2_ring(L, sh, masked, lds_limit, &wgs);
  return wgs;
}

hipError_t launch_fused2(const NumCols &num, const CatCols &cat, uint64_t rows, const CatLayout &L,
                         const CatDevice &D, int grid, size_t lds_limit, double *partials, unsigned *pair_slabs,
                         unsigned *skip, double *acc, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
                         const uint8_t *mask, unsigned long long *kept) {
  if (rows == 0) return hipSuccess;
  F2Shape sh;
  if (!f2_shape(L, sh)) return hipErrorInvalidValue;
  const bool masked = mask != nullptr;
  const int ring = f2_ring(L, sh, masked, lds_limit);
  if (ring < 2) return hipErrorInvalidValue;
  const F2Carve cv = f2_carve(L, sh, masked, ring);
  hipError_t e = hipErrorInvalidValue;
  if (ev0 && (e = hipEventRecord(ev0, stream)) != hipSuccess) return e;
#define GO(NBLK_, NBB_, MODE_) e = f2_launch_m<NBLK_, NBB_, MODE_>(sh.me, num, cat, rows, L, D, cv, ring, grid, partials, pair_slabs, skip, mask, kept, stream)
  constexpr int FULL = F2_PAIRS | F2_SSUM;
  if (sh.mode == 0) {                                         // NB
    switch (sh.nblk) { case 0: GO(0, 1, 0); break; case 1: GO(1, 1, 0); break; case 2: GO(2, 1, 0); break;
                       case 3: GO(3, 1, 0); break; case 4: GO(4, 1, 0); break; case 5: GO(5, 1, 0); break; default: break; }
  } else if (sh.mode == F2_PAIRS) {                           // n = 0
    GO(0, 1, F2_PAIRS);
  } else {                                                    // (nblk, nbb) pairs that exist for n = 1..20
    const int key = 10 * sh.nblk + sh.nbb;
    switch (key) {
      case 11: GO(1, 1, FULL); break; case 21: GO(2, 1, FULL); break; case 22: GO(2, 2, FULL); break;
      case 32: GO(3, 2, FULL); break; case 33: GO(3, 3, FULL); break; case 43: GO(4, 3, FULL); break;
      case 44: GO(4, 4, FULL); break; case 54: GO(5, 4, FULL); break; default: break;
    }
  }
#undef GO
  if (e != hipSuccess) return e;
  if (ev1 && (e = hipEventRecord(ev1, stream)) != hipSuccess) return e;
  if (sh.mode & F2_PAIRS) {
    hipLaunchKernelGGL(fused_pairs_fold2_kernel, dim3((L.n_p + 255) / 256), dim3(256), 0, stream, pair_slabs, grid,
                       L.n_p, D.p);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  if (L.n > 0) return launch_gram_fold(partials, grid, acc, stream);
  return hipSuccess;
}

namespace {
// layout of a group: n_sub numeric and m_sub key columns, 16 codes each, dictionaries where the
// aggregate's are
CatLayout sub_layout(const CatLayout &L, int n_sub, const int *cat_idx, int m_sub) {
  CatLayout S{};
  S.n = n_sub; S.m = m_sub; S.kind = 0;
  for (int c = 0; c < m_sub; c++) {
    const int g = cat_idx ? cat_idx[c] : c;
    S.ht_cap[c] = L.ht_cap[g]; S.ht_off[c] = L.ht_off[g]; S.kc[c] = 16;
    S.cnt_off[c] = 16 * c; S.s_off[c] = 16 * n_sub * c;
  }
  S.n_slots = L.n_slots;                          // (the kernel copies all dictionaries to LDS)
  S.n_cnt = 16 * m_sub; S.n_s = 16 * n_sub * m_sub; S.n_p = 0;
  return S;
}
bool sub_shape(int n_sub, int m_sub, F2Shape &sh) {
  if (n_sub < 1 || n_sub > 10 || m_sub < 1 || m_sub > 10) return false;
  sh.nblk = (n_sub + 3) / 4;
  sh.me = (m_sub + 1) / 2 * 2;
  sh.mode = F2_SSUM | F2_SUB;
  sh.nbb = (3 * n_sub + 1 + 15) / 16;
  return sh.me * sh.nbb <= F2_MAX_SBLOCKS;
}
}  // namespace

int fused2_sub_wgs_per_cu(int n_sub, int m_sub, bool masked, const CatLayout &L, size_t lds_limit) {
  F2Shape sh;
  int wgs = 1;
  if (!sub_shape(n_sub, m_sub, sh)) return 1;
  (void)f2_ring(sub_layout(L, n_sub, nullptr, m_sub), sh, masked, lds_limit, &wgs);
  return wgs;
}

bool fused2_sub_fits(int n_sub, int m_sub, bool masked, const CatLayout &L, size_t lds_limit) {
  F2Shape sh;
  if (!sub_shape(n_sub, m_sub, sh)) return false;
  const CatLayout S = sub_layout(L, n_sub, nullptr, m_sub);
  return f2_ring(S, sh, masked, lds_limit) >= 2;
}

hipError_t launch_fused2_sub(const NumCols &num, const CatCols &cat, uint64_t rows, const CatLayout &L,
                             const CatDevice &D, int k0, int n_sub, const int *cat_idx, int m_sub, bool do_cnt,
                             int grid, size_t lds_limit, const uint8_t *mask, hipStream_t stream) {
  if (rows == 0) return hipSuccess;
  F2Shape sh;
  if (!sub_shape(n_sub, m_sub, sh) || rows % TR != 0) return hipErrorInvalidValue;
  const CatLayout S = sub_layout(L, n_sub, cat_idx, m_sub);
  const bool masked = mask != nullptr;
  const int ring = f2_ring(S, sh, masked, lds_limit);
  if (ring < 2) return hipErrorInvalidValue;
  const F2Carve cv = f2_carve(S, sh, masked, ring);
  NumCols ns{};
  CatCols cs{};
  F2Sub sub{};
  for (int k = 0; k < n_sub; k++) ns.p[k] = num.p[k0 + k];
  for (int c = 0; c < m_sub; c++) {
    cs.p[c] = cat.p[cat_idx[c]];
    sub.s_goff[c] = L.s_off[cat_idx[c]];
    sub.cnt_goff[c] = L.cnt_off[cat_idx[c]];
  }
  sub.n_full = L.n; sub.k0 = k0; sub.do_cnt = do_cnt ? 1 : 0;
  hipError_t e = hipErrorInvalidValue;
#define GO(NBLK_, NBB_) e = f2_launch_m<NBLK_, NBB_, F2_SSUM | F2_SUB>(sh.me, ns, cs, rows, S, D, cv, ring, grid, nullptr, nullptr, nullptr, mask, nullptr, stream, sub)
  switch (10 * sh.nblk + sh.nbb) {
    case 11: GO(1, 1); break; case 21: GO(2, 1); break; case 22: GO(2, 2); break; case 32: GO(3, 2); break;
    default: break;
  }
#undef GO
  return e;
}

hipError_t launch_pairs_fold2(const unsigned *slabs, int nwg, int n_p, unsigned long long *p, hipStream_t stream) {
  hipLaunchKernelGGL(fused_pairs_fold2_kernel, dim3((n_p + 255) / 256), dim3(256), 0, stream, slabs, nwg, n_p, p);
  return hipGetLastError();
}

hipError_t launch_gather_units(const NumCols &num, const CatCols &cat, int n, int m, int unit, const unsigned *list,
                               unsigned count, unsigned *temp, uint64_t temp_stride, hipStream_t stream,
                               const uint8_t *mask, uint8_t *temp_mask) {
  if (count == 0) return hipSuccess;
  hipLaunchKernelGGL(gather_units_kernel, dim3(count < 8192 ? count : 8192), dim3(64), 0, stream, num, cat, n, m,
                     unit, list, count, temp, temp_stride, mask, temp_mask);
  return hipGetLastError();
}

}  // namespace cofactor
