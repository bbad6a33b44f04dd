// sum_to_triple_n_m with low-cardinality key columns (every column <= 16 distinct keys), triple
// kind, n >= 1: ONE pass over the n float and m int32 columns, as fused2.hip, but with the work of
// a 256-row tile split over SPECIALISED waves, two per SIMD.
//
// Reference loops replaced: duckdb_extension/src/triple/sum/sum_no_lift.cpp:119-214.
//
// Why (DESIGN.md "fused3_kernel"): fused2_kernel keeps the 45 pair-count tiles of m = 10 (180
// accumulator registers) next to the per-key-sum tiles (80) in every wave, so a wave needs the
// whole register file and runs alone on its SIMD: every LDS round trip and every MFMA dependency
// is exposed (5.1 ms per 1e8 rows at 10_10).  Here one 512-thread workgroup per CU runs
//
//   * 4 PAIR waves (waves 0-3): keys -> code bytes of the NEXT tile (byte table in LDS, hash probe
//     for keys outside 0..255), int8 one-hot operands, the pair counts of all column pairs on
//     v_mfma_i32_16x16x64_i8 (accumulators tied in the AGPRs); they also issue all LDS-DMA;
//   * 4 SUM waves (waves 4-7): floats -> three exact bf16 pieces, the dense Gram on
//     v_mfma_f32_4x4x1 straight from the raw tile, and the per-key sums / key counts as
//     onehot^T [pieces | 1] on v_mfma_f32_16x16x32_bf16;
//
// wave w and wave w + 4 share a SIMD and the same 64 rows of every tile, so one wave's MFMAs run
// beside the other's VALU work, and both fit 256 registers.  The code bytes are the only data
// that crosses waves: the pair wave writes the codes of tile t + 1 while everybody works on tile
// t (two code buffers), so ONE s_barrier per tile orders everything; the raw ring therefore holds
// the tile in use, the tile the pair waves translate, and R - 2 tiles in flight.
#include <cstdio>
#include <cstdlib>

#include "onepass.hpp"

namespace cofactor {
using namespace onepass;

namespace {

constexpr int F3_THREADS = 512;
constexpr int CODE_COL = TR + 16;          // bytes of one code column of a code buffer (256 rows + bank spread)

#ifdef F3_STAMPS
#define F3_STAMP(var) do { if (ablate & 64) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); var += now_ - stamp_; stamp_ = now_; } } while (0)
#else
#define F3_STAMP(var) do { } while (0)
#endif

// MFMAs and operand arithmetic as builtins / plain C: a kernel that names no AGPR gets VGPR-form
// MFMAs from hipcc, which then knows every latency and hazard and can schedule operand
// preparation between the matrix instructions (the asm forms of onepass.hpp are opaque to it:
// measured 2.0 ms for the per-key-sum phase alone with them, LDS reads issued right before use).
__device__ __forceinline__ void f3_pmfma(i32x4 &acc, i32x4 a, i32x4 b) {
#if defined(F3_ASM_MFMA) || !defined(F3_BUILTIN_PAIRS)
  pmfma_v(acc, a, b);
#else
  acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
#endif
}
__device__ __forceinline__ void f3_smfma(f32x4 &acc, u32x4 a, u32x4 b) {
#ifdef F3_ASM_MFMA
  smfma(acc, a, b);
#else
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
#endif
}
__device__ __forceinline__ unsigned f3_onehot_i8(unsigned codes4, unsigned ixor) {   // 4 code bytes -> 0x40 where code == i
#ifdef F3_ASM_MFMA
  return xad(codes4, ixor, 0x21212121u) & 0x40404040u;
#else
  return ((codes4 ^ ixor) + 0x21212121u) & 0x40404040u;
#endif
}
// int8 one-hot bytes (0x40) of 8 rows -> bf16 one-hot operand (0x4000 = 2.0)
__device__ __forceinline__ u32x4 f3_onehot_bf16(unsigned w0, unsigned w1) {
#ifdef F3_ASM_MFMA
  return onehot_bf16(w0, w1);
#else
  u32x4 r;
  r[0] = __builtin_amdgcn_perm(0u, w0, 0x010C000Cu); r[1] = __builtin_amdgcn_perm(0u, w0, 0x030C020Cu);
  r[2] = __builtin_amdgcn_perm(0u, w1, 0x010C000Cu); r[3] = __builtin_amdgcn_perm(0u, w1, 0x030C020Cu);
  return r;
#endif
}

struct F3Carve {          // byte offsets into the dynamic LDS block
  int ring, slot_bytes, zero, codes, codes_buf, drop, pieces, pieces_bytes, s, cnt, direct, slot, dcode, total;
};

template <int NBLK, int NBB, int M>
__global__ __launch_bounds__(F3_THREADS) void fused3_kernel(NumCols num, CatCols cat, uint64_t rows, CatLayout L,
                                                            CatDevice D, F3Carve cv, int ring,
                                                            double *__restrict__ partials,
                                                            unsigned *__restrict__ pair_slabs,
                                                            unsigned *__restrict__ skip,
                                                            const uint8_t *__restrict__ mask,
                                                            unsigned long long *__restrict__ kept, int ablate) {
#ifndef COFACTOR_DEV_ABLATE
  ablate = 0;                                                // (dev builds only: tests/tools/f3_ablate.py)
#endif
  constexpr int NPAIR = NBLK * (NBLK + 1) / 2;
  constexpr int NT = M * (M - 1) / 2;                        // 16x16 pair blocks (c1 < c2)
  // the last NSP blocks, (M-3, M-2), (M-3, M-1), (M-2, M-1), are accumulated by the SUM wave of the
  // same rows (it forms those one-hots anyway): 12 registers less in the pair waves, which sit at
  // the 256-register limit (a single spill there drains the DMA ring, see make_codes)
  constexpr int NSP = M >= 4 ? 3 : 0;
  constexpr int NTP = NT - NSP;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool sum_role = wave >= 4;                           // waves 4..7: Gram + per-key sums
  const int sw = wave & 3;                                   // the 64 rows of a tile this wave works on
  const int n = L.n, m = L.m;                                // m <= M (M is m rounded up to even)
  const bool masked = mask != nullptr;
  const int ndata = n + m;                                   // DMA'd 1-KiB columns per tile
  const int ncols = ndata + (masked ? 1 : 0);                // + the row filter (256 bytes)
  const int pcols = 3 * n;                                   // piece columns; column `pcols` is the ones column

  double *l_s = reinterpret_cast<double *>(lds + cv.s);
  unsigned *l_cnt = reinterpret_cast<unsigned *>(lds + cv.cnt);
  unsigned char *l_direct = lds + cv.direct;
  unsigned char *l_far = l_direct + M * DIRECT_STRIDE;
  int *l_hoff = reinterpret_cast<int *>(l_far + 32);         // per column: first dictionary slot, slots (for the probe)
  int *l_hcap = l_hoff + 12;
  unsigned long long *l_slot = reinterpret_cast<unsigned long long *>(lds + cv.slot);
  int32_t *l_dcode = reinterpret_cast<int32_t *>(lds + cv.dcode);
  unsigned char *l_codes = lds + cv.codes;                   // [2][M][CODE_COL]
  unsigned *l_drop = reinterpret_cast<unsigned *>(lds + cv.drop);   // [2][4]: the 64-row block was left out
  unsigned char *my_pieces = lds + cv.pieces + sw * cv.pieces_bytes;   // (sum waves) [16 * NBB][PST]

  // ---- one-time LDS setup ------------------------------------------------------------------
  for (int i = tid; i < (cv.total - cv.zero) / 4; i += F3_THREADS) reinterpret_cast<unsigned *>(lds + cv.zero)[i] = 0u;
  __syncthreads();
  for (int i = tid; i < L.n_slots; i += F3_THREADS) { l_slot[i] = D.ht_slot[i]; l_dcode[i] = D.ht_code[i]; }
  for (int i = tid; i < M * DIRECT_STRIDE; i += F3_THREADS) l_direct[i] = (unsigned char)NO_CODE;
  for (int i = tid; i < 2 * cv.codes_buf; i += F3_THREADS) l_codes[i] = (unsigned char)NO_CODE;   // (odd m: column M - 1 for ever)
  if (tid < m) { l_hoff[tid] = L.ht_off[tid]; l_hcap[tid] = L.ht_cap[tid]; }
  if (sum_role) {                                            // ones column: bf16 1.0
    unsigned short *ones = reinterpret_cast<unsigned short *>(my_pieces + pcols * PST);
    ones[lane] = 0x3F80;
  }
  __syncthreads();
  for (int c = 0; c < m; c++)
    for (int i = tid; i < L.ht_cap[c]; i += F3_THREADS) {
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
  {
    const int b = lane >> 2, t = lane & 3;
    if (b < RPM * NPAIR) {
      rsub = b / NPAIR;
      int bi = 0, rem = b % NPAIR;
      while (rem >= NBLK - bi) { rem -= NBLK - bi; bi++; }
      colA = 4 * bi + t;
      colB = 4 * (bi + rem) + t;
    }
  }
  const bool okA = colA >= 0 && colA < n, okB = colB >= 0 && colB < n;
  const int g_row = (sw * 64 + 4 * rsub) * 4;
  const int offA = colA * COLB + g_row, offB = colB * COLB + g_row;
  // one-hot operands (16x16 MFMAs): lane (i = lane & 15, q = lane >> 4) holds code value i for the
  // rows 16 q .. 16 q + 15 of the wave's 64 rows.  (code ^ i ^ 31) is 31 exactly on a match, and
  // adding 0x21 carries into bit 6 exactly then (codes are 0..17: no carry between the bytes).
  const int li = lane & 15, lq = lane >> 4;
  const unsigned ixor = (unsigned)(li ^ 31) * 0x01010101u;

  const uint64_t ntiles = rows / TR;
  const uint64_t G = gridDim.x;

  // ---- tile ring: the pair waves issue every LDS-DMA; wave w loads the virtual columns w, w + 4, .. ---
  const unsigned lds0 = (unsigned)(unsigned long long)(lds_void *)lds;   // LDS byte address of the block
  constexpr int MAXCPW = (4 * NBLK + M + 1 + 3) / 4;
  const int mycnt = sum_role ? 0 : (ncols - wave + 3) / 4;   // DMA instructions of this wave per tile
  const unsigned char *dsrc[MAXCPW];
  unsigned doff[MAXCPW];
#pragma unroll
  for (int i = 0; i < MAXCPW; i++) {
    const int vc = min(sw + 4 * i, ncols - 1);
    dsrc[i] = vc < n ? reinterpret_cast<const unsigned char *>(num.p[min(vc, COFACTOR_MAX_NUM - 1)])
                     : (vc < ndata ? reinterpret_cast<const unsigned char *>(cat.p[min(max(vc - n, 0), COFACTOR_MAX_CAT - 1)])
                                   : reinterpret_cast<const unsigned char *>(mask));
    doff[i] = (unsigned)(vc * COLB);
  }
  const unsigned lane16 = 16u * lane, lane4 = 4u * lane;
  auto dma_tile = [&](uint64_t t, int slot) {
    const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + cv.ring + slot * cv.slot_bytes);
#pragma unroll
    for (int i = 0; i < MAXCPW; i++)
      if (i < mycnt) {
        if (!masked || sw + 4 * i < ndata)
          glds16_s(dsrc[i] + t * (TR * 4), lane16, __builtin_amdgcn_readfirstlane(base + doff[i]));
        else
          glds4_s(dsrc[i] + t * TR, lane4, __builtin_amdgcn_readfirstlane(base + doff[i]));
      }
  };

  const int keep = (ring - 3) * mycnt;                       // DMA instructions that may stay in flight at a wait
  unsigned *l_p = reinterpret_cast<unsigned *>(lds + cv.ring);            // (end: the ring holds the pair table ...
  double *red = reinterpret_cast<double *>(lds + cv.ring) + (L.n_p + 1) / 2;   //  ... and the 4 Gram images)
  __syncthreads();                                           // LDS setup visible

  // The two roles run their own loops (their accumulators must not be live in each other's code:
  // 180 + 96 registers do not fit 256), meeting at the same sequence of barriers: one after the
  // prologue's wait, one per tile, three at the end.
  if (!sum_role) {
    // =================================== PAIR waves ====================================================
    i32x4 pacc[NTP];
#pragma unroll
    for (int q = 0; q < NTP; q++) pacc[q] = i32x4{0, 0, 0, 0};
    unsigned n_kept = 0;
    // keys of this wave's 64 rows of the tile in `slot` -> code bytes in code buffer `cbuf`
    auto make_codes = [&](uint64_t t, int slot, int cbuf) {
      unsigned char *base = lds + cv.ring + slot * cv.slot_bytes;
      unsigned char *cdst = l_codes + cbuf * cv.codes_buf + sw * 64;
      const unsigned char *mrow = base + ndata * COLB + sw * 64;            // this wave's 64 filter bytes
      const int q4 = 4 * (lane & 15);                                       // this lane's 4 rows
      unsigned fl = 0x01010101u;
      if (masked) fl = *reinterpret_cast<const unsigned *>(mrow + q4);
      bool unknown = false;
      // Batched: all key reads, then all table lookups, then the packing — two exposed LDS round
      // trips per tile instead of two per column group (a lone wave hides no latency).
      constexpr int NJ = (M + 3) / 4;
      int cj[NJ];
      uint4 kv[NJ];
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        cj[j] = min(4 * j + (lane >> 4), m - 1);                            // lanes past the last column redo it
        kv[j] = *reinterpret_cast<const uint4 *>(base + (n + cj[j]) * COLB + (sw * 64 + q4) * 4);
      }
      unsigned cd[NJ][4];
      unsigned farc[NJ];
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const unsigned char *dt = l_direct + cj[j] * DIRECT_STRIDE;         // byte table: keys 0..255, [256] = NO_CODE
        cd[j][0] = dt[min(kv[j].x, (unsigned)DIRECT_KEYS)]; cd[j][1] = dt[min(kv[j].y, (unsigned)DIRECT_KEYS)];
        cd[j][2] = dt[min(kv[j].z, (unsigned)DIRECT_KEYS)]; cd[j][3] = dt[min(kv[j].w, (unsigned)DIRECT_KEYS)];
        farc[j] = l_far[cj[j]];
      }
      bool probe = false;
#pragma unroll
      for (int j = 0; j < NJ; j++)
        probe = probe || (farc[j] && ((cd[j][0] | cd[j][1] | cd[j][2] | cd[j][3]) & NO_CODE));
      if (__builtin_amdgcn_ballot_w64(probe) != 0ull) {                     // a column holds keys outside 0..255: probe
#pragma unroll
        for (int j = 0; j < NJ; j++) {                                      // (unrolled: a runtime index would put kv / cd on the stack)
          if (!(farc[j] && ((cd[j][0] | cd[j][1] | cd[j][2] | cd[j][3]) & NO_CODE))) continue;
          const unsigned long long *sl = l_slot + l_hoff[cj[j]];
          const int32_t *dc = l_dcode + l_hoff[cj[j]];
          const int cap = l_hcap[cj[j]];
          const unsigned kk[4] = {kv[j].x, kv[j].y, kv[j].z, kv[j].w};
#pragma unroll
          for (int e = 0; e < 4; e++)
            if (cd[j][e] == NO_CODE) cd[j][e] = lds_lookup1(sl, dc, cap, kk[e]);
        }
      }
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        unsigned cx = cd[j][0], cy = cd[j][1], cz = cd[j][2], cw = cd[j][3];
        cx = (fl & 0x000000FFu) ? cx : ROW_OFF; cy = (fl & 0x0000FF00u) ? cy : ROW_OFF;
        cz = (fl & 0x00FF0000u) ? cz : ROW_OFF; cw = (fl & 0xFF000000u) ? cw : ROW_OFF;
        unknown = unknown || cx == NO_CODE || cy == NO_CODE || cz == NO_CODE || cw == NO_CODE;
        *reinterpret_cast<unsigned *>(cdst + cj[j] * CODE_COL + q4) = cx | (cy << 8) | (cz << 16) | (cw << 24);
      }
      unsigned dropped = 0u;
      if (__builtin_amdgcn_ballot_w64(unknown) != 0ull) {
        // optimistic mode: the 64 rows are left out as a whole and redone by the host after a
        // dictionary pass; otherwise the dictionary pass has missed a key (reported at the next sync)
        if (lane == 0) {
          if (skip) skip[1 + atomicAdd(&skip[0], 1u)] = (unsigned)(t * 4 + sw);
          else D.flags[1] = 1;
        }
        wait_vmcnt_imm<0>();                                   // (rare path: keep the hand-counted waits exact)
        fl = 0u;
        dropped = 1u;
        for (int c0 = 0; c0 < m; c0 += 4) {
          const int c = min(c0 + (lane >> 4), m - 1);
          *reinterpret_cast<unsigned *>(cdst + c * CODE_COL + q4) = ROW_OFF * 0x01010101u;
        }
      }
      if (lane == 0) l_drop[4 * cbuf + sw] = dropped;
      if (masked && lane < 16)                                 // lanes 0..15 hold the 64 filter bytes once
        n_kept += ((fl & 0x000000FFu) != 0) + ((fl & 0x0000FF00u) != 0) + ((fl & 0x00FF0000u) != 0) + ((fl & 0xFF000000u) != 0);
    };
    // pair counts of this wave's 64 rows from the code bytes in buffer `cbuf`
    auto pair_products = [&](int cbuf) {
      const unsigned char *csrc = l_codes + cbuf * cv.codes_buf + sw * 64 + 16 * lq;
      i32x4 oh[M];
#pragma unroll
      for (int c = 0; c < M; c++) {
        const uint4 cb = *reinterpret_cast<const uint4 *>(csrc + c * CODE_COL);
        oh[c][0] = (int)f3_onehot_i8(cb.x, ixor);
        oh[c][1] = (int)f3_onehot_i8(cb.y, ixor);
        oh[c][2] = (int)f3_onehot_i8(cb.z, ixor);
        oh[c][3] = (int)f3_onehot_i8(cb.w, ixor);
      }
#if defined(F3_ASM_MFMA) || !defined(F3_BUILTIN_PAIRS)
      settle_operands<M>(oh);
#endif
      int q = 0;
#pragma unroll
      for (int c1 = 0; c1 < M; c1++)
#pragma unroll
        for (int c2 = c1 + 1; c2 < M; c2++, q++)
          if (q < NTP) f3_pmfma(pacc[q], oh[c1], oh[c2]);
    };


    uint64_t t = blockIdx.x;                                 // (blockIdx.x < ntiles: the launcher sizes the grid)
    for (int r = 0; r < ring - 1; r++) dma_tile(min(t + r * G, ntiles - 1), r);
    wait_vmcnt(keep);                                        // tiles 0 and 1 of this workgroup have landed
    __builtin_amdgcn_s_barrier();
    make_codes(t, 0, 0);
    int slot = 0, cbuf = 0;
    [[maybe_unused]] unsigned long long stamp_ = 0, c_wait = 0, c_bar = 0, c_dma = 0, c_prod = 0, c_codes = 0;
#ifdef F3_STAMPS
    stamp_ = __builtin_amdgcn_s_memtime();
#endif
    for (; t < ntiles; t += G) {
      wait_vmcnt(keep);                                      // this wave's part of tile t + G has landed ...
      F3_STAMP(c_wait);
      __builtin_amdgcn_s_barrier();                          // ... and everybody else's; slot - 1 is free
      F3_STAMP(c_bar);
      int nslot = slot + ring - 1;
      nslot = nslot >= ring ? nslot - ring : nslot;
      const int slot1 = slot + 1 == ring ? 0 : slot + 1;
      dma_tile(min(t + (uint64_t)(ring - 1) * G, ntiles - 1), nslot);     // past the end: a harmless re-load
      F3_STAMP(c_dma);
      if (!(ablate & 1)) pair_products(cbuf);
      F3_STAMP(c_prod);
      if (t + G < ntiles && !(ablate & 2)) make_codes(t + G, slot1, cbuf ^ 1);
      F3_STAMP(c_codes);
      slot = slot1;
      cbuf ^= 1;
    }
#ifdef F3_STAMPS
    if ((ablate & 64) && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 131))
      printf("wg %3d pair wave %d: wait %llu barrier %llu dma %llu products %llu codes %llu  (cycles, %llu tiles)\n", (int)blockIdx.x, wave,
             c_wait, c_bar, c_dma, c_prod, c_codes, (unsigned long long)((ntiles - blockIdx.x + G - 1) / G));
#endif
    wait_vmcnt_imm<0>();                                     // drain the re-loads before the ring is reused
    __syncthreads();
    for (int i = tid; i < L.n_p; i += F3_THREADS) l_p[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NTP; q++) {
#if defined(F3_ASM_MFMA) || !defined(F3_BUILTIN_PAIRS)
      mfma_settle_v(pacc[q]);
#endif
    }
    int q = 0;
#pragma unroll
    for (int c1 = 0; c1 < M; c1++)
#pragma unroll
      for (int c2 = c1 + 1; c2 < M; c2++, q++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const unsigned v = q < NTP ? (unsigned)pacc[q][r] >> 12 : 0u;   // the one-hots are 64: 64 x 64 per row
          if (c2 < m && v) {
            const int qi = c1 * m - c1 * (c1 - 1) / 2 + (c2 - c1);
            atomicAdd(&l_p[256 * qi + 16 * (4 * lq + r) + li], v);
          }
        }
    if (masked) {
      unsigned long long kk = n_kept;
      for (int off = 32; off > 0; off >>= 1) kk += __shfl_down(kk, off, 64);
      if (lane == 0 && kk && kept) atomicAdd(kept, kk);
    }
    __syncthreads();
  } else {
    // =================================== SUM waves =====================================================
    f32x4 sacc[M][NBB];
#pragma unroll
    for (int c = 0; c < M; c++)
#pragma unroll
      for (int bb = 0; bb < NBB; bb++) sacc[c][bb] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    f32x2 ls_lo = {0.f, 0.f}, ls_hi = {0.f, 0.f};
    double dq0 = 0, dq1 = 0, dq2 = 0, dq3 = 0, dl = 0;
    i32x4 spacc[NSP > 0 ? NSP : 1];                          // the pair blocks this wave accumulates
#pragma unroll
    for (int q = 0; q < (NSP > 0 ? NSP : 1); q++) spacc[q] = i32x4{0, 0, 0, 0};
    [[maybe_unused]] unsigned long long stamp_ = 0, c_bar = 0, c_pieces = 0, c_gram = 0, c_s = 0, c_flush = 0;
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
        for (int bb = 0; bb < NBB; bb++) {
#ifdef F3_ASM_MFMA
          mfma_settle(sacc[c][bb]);
#endif
        }
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

    // this wave's 64 rows of the tile in `slot`, code bytes in buffer `cbuf`
    auto sum_subtile = [&](int slot, int cbuf) {
      unsigned char *base = lds + cv.ring + slot * cv.slot_bytes;
      const unsigned char *mrow = base + ndata * COLB + sw * 64;
      const int q4 = 4 * (lane & 15);
      unsigned fl = 0x01010101u;
      if (masked) fl = *reinterpret_cast<const unsigned *>(mrow + q4);
      if (l_drop[4 * cbuf + sw]) fl = 0u;                                   // the pair wave left the block out
      const bool some_dropped = __builtin_amdgcn_ballot_w64(fl != 0x01010101u) != 0ull;   // wave-uniform
      // -- floats -> bf16 pieces; dropped rows become zeros in the raw tile too (the Gram reads it) --
      bool nonfinite = false;
      // (a lone wave hides no LDS latency: every read of a phase is issued before its first use)
      const unsigned char *csrc = l_codes + cbuf * cv.codes_buf + sw * 64;
      uint4 xin[NBLK];
#pragma unroll
      for (int j = 0; j < NBLK; j++)
        xin[j] = *reinterpret_cast<const uint4 *>(base + min(4 * j + (lane >> 4), n - 1) * COLB + (sw * 64 + q4) * 4);
      __builtin_amdgcn_sched_barrier(0);
      if (!(ablate & 4))
#pragma unroll
      for (int j = 0; j < NBLK; j++) {
        const int c = min(4 * j + (lane >> 4), n - 1);                      // lanes past the last column redo it
        uint4 *src = reinterpret_cast<uint4 *>(base + c * COLB + (sw * 64 + q4) * 4);
        const uint4 xv = xin[j];
        unsigned u[4] = {xv.x, xv.y, xv.z, xv.w};
        if (some_dropped) {
          u[0] = (fl & 0x000000FFu) ? u[0] : 0u; u[1] = (fl & 0x0000FF00u) ? u[1] : 0u;
          u[2] = (fl & 0x00FF0000u) ? u[2] : 0u; u[3] = (fl & 0xFF000000u) ? u[3] : 0u;
          *src = make_uint4(u[0], u[1], u[2], u[3]);
        }
        // x = hi + mid + lo, each a bf16 (exact): hi = upper half of x, mid = upper half of
        // x - hi, lo = x - hi - mid (its lower half is zero)
        constexpr unsigned UPPER_HALVES = 0x07060302u;                      // {s0.b3, s0.b2, s1.b3, s1.b2}
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
      // the prep stores of this wave are read by other lanes of the SAME wave below: LDS executes a
      // wave's instructions in order; this only keeps the compiler from moving loads above the stores
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (__builtin_amdgcn_ballot_w64(nonfinite) != 0ull) {
        // rare: add every inf / nan of these 64 rows straight to its keys' cells
        for (int k = 0; k < n; k++) {
          const float x = *reinterpret_cast<const float *>(base + k * COLB + (sw * 64 + lane) * 4);
          if ((__float_as_uint(x) & 0x7F800000u) == 0x7F800000u)
            for (int c = 0; c < m; c++) {
              const unsigned cd = csrc[c * CODE_COL + lane];
              if (cd < 16u) unsafeAtomicAdd(&l_s[L.s_off[c] + (int)cd * n + k], (double)x);
            }
        }
      }
      F3_STAMP(c_pieces);
      // -- the dense Gram straight from the raw tile --
      u32x4 bop[2][NBB];
#pragma unroll
      for (int s = 0; s < 2; s++)
#pragma unroll
        for (int bb = 0; bb < NBB; bb++)
          bop[s][bb] = *reinterpret_cast<const u32x4 *>(my_pieces + (16 * bb + li) * PST + 2 * (16 * lq + 8 * s));
      if (!(ablate & 8)) {
        const f32x4 *va = reinterpret_cast<const f32x4 *>(okA ? base + offA : lds + cv.zero + g_row);
        const f32x4 *vb = reinterpret_cast<const f32x4 *>(okB ? base + offB : lds + cv.zero + g_row);
        constexpr int GIT = 16 / RPM, GB = GIT > 4 ? 4 : GIT;   // operand reads in batches of GB row groups
#pragma unroll
        for (int it0 = 0; it0 < GIT; it0 += GB) {
        f32x4 ga[GB], gb[GB];
#pragma unroll
        for (int it = 0; it < GB; it++) { ga[it] = va[(it0 + it) * RPM]; gb[it] = vb[(it0 + it) * RPM]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < GB; it++) {
          const f32x4 a = ga[it], bv = gb[it];
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], bv[0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], bv[1], acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], bv[2], acc2, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], bv[3], acc3, 0, 0, 0);
          ls_lo += __builtin_shufflevector(a, a, 0, 1);
          ls_hi += __builtin_shufflevector(a, a, 2, 3);
        }
        }
      }
      // -- per-key sums and key counts: onehot(code)^T [pieces | 1] over the wave's 64 rows.  k index of
      //    the 16x16x32 MFMA: lane (li, lq), element j <-> row 16 lq + 8 s + j for the s-th instruction
      //    (A and B are built the same way, so any k order of the instruction pairs the same rows) --
      F3_STAMP(c_gram);
      if (ablate & 16) return;
      uint4 cbq[M];
#pragma unroll
      for (int c = 0; c < M; c++) cbq[c] = *reinterpret_cast<const uint4 *>(csrc + c * CODE_COL + 16 * lq);
      __builtin_amdgcn_sched_barrier(0);
      i32x4 ohk[NSP > 0 ? NSP : 1];
#pragma unroll
      for (int c = 0; c < M; c++) {
        const uint4 cb = cbq[c];
        const unsigned o0 = f3_onehot_i8(cb.x, ixor), o1 = f3_onehot_i8(cb.y, ixor);
        const unsigned o2 = f3_onehot_i8(cb.z, ixor), o3 = f3_onehot_i8(cb.w, ixor);
        if (NSP > 0 && c >= M - NSP) ohk[c - (M - NSP)] = i32x4{(int)o0, (int)o1, (int)o2, (int)o3};
        const u32x4 ab0 = f3_onehot_bf16(o0, o1);
        const u32x4 ab1 = f3_onehot_bf16(o2, o3);
#pragma unroll
        for (int bb = 0; bb < NBB; bb++) f3_smfma(sacc[c][bb], ab0, bop[0][bb]);
#pragma unroll
        for (int bb = 0; bb < NBB; bb++) f3_smfma(sacc[c][bb], ab1, bop[1][bb]);
      }
      if (NSP > 0) {
        asm volatile("s_nop 3" : "+v"(ohk[0]), "+v"(ohk[NSP > 1 ? 1 : 0]), "+v"(ohk[NSP > 2 ? 2 : 0]));
        pmfma_v(spacc[0], ohk[0], ohk[NSP > 1 ? 1 : 0]);
        pmfma_v(spacc[NSP > 1 ? 1 : 0], ohk[0], ohk[NSP > 2 ? 2 : 0]);
        pmfma_v(spacc[NSP > 2 ? 2 : 0], ohk[NSP > 1 ? 1 : 0], ohk[NSP > 2 ? 2 : 0]);
      }
      F3_STAMP(c_s);
    };


    __builtin_amdgcn_s_barrier();                            // (the pair waves' prologue wait)
    int slot = 0, cbuf = 0, since_g = 0, since_s = 0;
#ifdef F3_STAMPS
    stamp_ = __builtin_amdgcn_s_memtime();
#endif
    for (uint64_t t = blockIdx.x; t < ntiles; t += G) {
      __builtin_amdgcn_s_barrier();                          // tile t is in `slot`, its code bytes in buffer `cbuf`
      F3_STAMP(c_bar);
      if (!(ablate & 32)) sum_subtile(slot, cbuf);
      if (++since_g == G_FLUSH_TILES * RPM) { flush_gram(); since_g = 0; }
      if (++since_s == S_FLUSH_TILES) { flush_s(); since_s = 0; }
      F3_STAMP(c_flush);
      slot = slot + 1 == ring ? 0 : slot + 1;
      cbuf ^= 1;
    }
#ifdef F3_STAMPS
    if ((ablate & 64) && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 131))
      printf("wg %3d sum  wave %d: barrier %llu pieces %llu gram %llu S %llu flush %llu\n", (int)blockIdx.x, wave,
             c_bar, c_pieces, c_gram, c_s, c_flush);
#endif
    flush_gram();
    flush_s();
    __syncthreads();
    for (int i = tid; i < L.n_p; i += F3_THREADS) l_p[i] = 0u;
    __syncthreads();
    if (NSP > 0) {
      const int cc[3][2] = {{M - 3, M - 2}, {M - 3, M - 1}, {M - 2, M - 1}};
#pragma unroll
      for (int q = 0; q < NSP; q++) {
        mfma_settle_v(spacc[q]);
        const int c1 = cc[q][0], c2 = cc[q][1];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const unsigned v = (unsigned)spacc[q][r] >> 12;
          if (c2 < m && v) {
            const int qi = c1 * m - c1 * (c1 - 1) / 2 + (c2 - c1);
            atomicAdd(&l_p[256 * qi + 16 * (4 * lq + r) + li], v);
          }
        }
      }
    }
    double *mine = red + sw * GRAM_ACC_LEN;
    mine[0 * 64 + lane] = dq0; mine[1 * 64 + lane] = dq1; mine[2 * 64 + lane] = dq2; mine[3 * 64 + lane] = dq3;
    mine[4 * 64 + lane] = dl;
    __syncthreads();
  }

  // ---- end: the workgroup's pair slab, Gram image, count / sum tables ------------------------------------
  {
    unsigned *slab = pair_slabs + (uint64_t)blockIdx.x * (uint64_t)L.n_p;
    for (int i = tid; i < L.n_p; i += F3_THREADS) slab[i] = l_p[i];
    for (int i = tid; i < GRAM_ACC_LEN; i += F3_THREADS) {   // 4 waves and RPM row groups, fixed order
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
  // key counts: to cnt and to the diagonal cells (k, k) of the column's own pair table
  for (int i = tid; i < 16 * m; i += F3_THREADS)
    if (l_cnt[i]) {
      atomicAdd(&D.cnt[i], (unsigned long long)l_cnt[i]);
      const int c = i >> 4, code = i & 15;
      const int qd = c * m - c * (c - 1) / 2;
      atomicAdd(&D.p[L.p_off[qd] + code * 16 + code], (unsigned long long)l_cnt[i]);
    }
  for (int i = tid; i < L.n_s; i += F3_THREADS)
    if (l_s[i] != 0.0) unsafeAtomicAdd(&D.s[i], l_s[i]);
}

struct F3Shape { int nblk, nbb, me; };     // me = m rounded up to even (template parameter M)

bool f3_shape(const CatLayout &L, F3Shape &sh) {
  if (L.kind != 0 || L.m < 2 || L.m > 10 || L.n < 1 || L.n > COFACTOR_MAX_NUM) return false;
  sh.nblk = (L.n + 3) / 4;
  sh.me = (L.m + 1) / 2 * 2;
  sh.nbb = (3 * L.n + 1 + 15) / 16;               // piece columns + the ones column, 16 per block
  // the sum waves' accumulators (per-key-sum blocks + Gram) must fit beside the pair waves' split
  return sh.me * sh.nbb * 4 + 16 <= 176;
}

F3Carve f3_carve(const CatLayout &L, const F3Shape &sh, bool masked, int ring) {
  F3Carve c{};
  size_t o = 0;
  auto take = [&](size_t bytes, size_t align) { o = (o + align - 1) / align * align; size_t at = o; o += bytes; return (int)at; };
  c.slot_bytes = (L.n + L.m) * COLB + (masked ? 256 : 0);
  c.slot_bytes = (c.slot_bytes + 15) / 16 * 16;
  const size_t ring_bytes = (size_t)ring * c.slot_bytes;
  // at the end the ring area holds the workgroup's pair table and the 4 Gram images
  const size_t tail_bytes = (size_t)(L.n_p + 1) / 2 * 8 + sizeof(double) * 4 * GRAM_ACC_LEN;
  c.ring = take(ring_bytes > tail_bytes ? ring_bytes : tail_bytes, 16);
  c.zero = take(COLB, 16);                        // (everything from here on is zeroed at start)
  c.codes_buf = sh.me * CODE_COL;
  c.codes = take((size_t)2 * c.codes_buf, 16);
  c.drop = take(8 * 4, 4);
  c.pieces_bytes = 16 * sh.nbb * PST;
  c.pieces = take((size_t)4 * c.pieces_bytes, 16);
  c.s = take((size_t)L.n_s * 8, 8);
  c.cnt = take((size_t)16 * L.m * 4, 4);
  c.direct = take((size_t)sh.me * DIRECT_STRIDE + 32 + 24 * 4, 4);
  c.slot = take((size_t)L.n_slots * 8, 8);
  c.dcode = take((size_t)L.n_slots * 4, 4);
  c.total = (int)((o + 15) / 16 * 16);
  return c;
}

// deepest ring (<= 8 slots, >= 4: one tile in use, one being translated, two in flight after an issue)
int f3_ring(const CatLayout &L, const F3Shape &sh, bool masked, size_t lds_limit) {
  static const long forced = [] { const char *v = getenv("COFACTOR_F3_RING"); return v ? atol(v) : 0l; }();
  int best = 0;
  for (int r = 4; r <= 8; r++) {
    const F3Carve c = f3_carve(L, sh, masked, r);
    const int cpw = (L.n + L.m + (masked ? 1 : 0) + 3) / 4;
    if ((size_t)c.total <= lds_limit && (r - 3) * cpw <= 48 && (forced == 0 || r <= forced)) best = r;
  }
  return best;
}

template <int NBLK, int NBB, int M>
hipError_t f3_launch_one(const NumCols &num, const CatCols &cat, uint64_t rows, const CatLayout &L,
                         const CatDevice &D, const F3Carve &cv, int ring, int grid, double *partials,
                         unsigned *slabs, unsigned *skip, const uint8_t *mask, unsigned long long *kept,
                         hipStream_t stream) {
  int ablate = 0;
#ifdef COFACTOR_DEV_ABLATE
  if (const char *v = getenv("COFACTOR_F3_ABLATE")) ablate = atoi(v);
#endif
  hipError_t e = hipFuncSetAttribute((const void *)fused3_kernel<NBLK, NBB, M>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, cv.total);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((fused3_kernel<NBLK, NBB, M>), dim3(grid), dim3(F3_THREADS), cv.total, stream,
                     num, cat, rows, L, D, cv, ring, partials, slabs, skip, mask, kept, ablate);
  return hipGetLastError();
}

template <int NBLK, int NBB>
hipError_t f3_launch_m(int me, const NumCols &num, const CatCols &cat, uint64_t rows, const CatLayout &L,
                       const CatDevice &D, const F3Carve &cv, int ring, int grid, double *partials,
                       unsigned *slabs, unsigned *skip, const uint8_t *mask, unsigned long long *kept,
                       hipStream_t stream) {
  switch (me) {
#define CASE(M_) case M_: if constexpr (M_ * NBB * 4 + 16 <= 176) \
      return f3_launch_one<NBLK, NBB, M_>(num, cat, rows, L, D, cv, ring, grid, partials, slabs, skip, mask, kept, stream); \
    else break;
#ifndef F3_DEV_ONLY_10_10
    CASE(2) CASE(4) CASE(6) CASE(8)
#endif
    CASE(10)
#undef CASE
    default: break;
  }
  return hipErrorInvalidValue;
}

}  // namespace

bool fused3_applicable(const CatLayout &L, const int32_t *nkeys, bool masked, size_t lds_limit) {
  for (int c = 0; c < L.m; c++)
    if (nkeys[c] > 16 || L.kc[c] != 16) return false;
  F3Shape sh;
  if (!f3_shape(L, sh)) return false;
  return f3_ring(L, sh, masked, lds_limit) >= 4;
}

hipError_t launch_fused3(const NumCols &num, const CatCols &cat, uint64_t rows, const CatLayout &L,
                         const CatDevice &D, int grid, size_t lds_limit, double *partials, unsigned *pair_slabs,
                         unsigned *skip, double *acc, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
                         const uint8_t *mask, unsigned long long *kept) {
  if (rows == 0) return hipSuccess;
  F3Shape sh;
  if (!f3_shape(L, sh)) return hipErrorInvalidValue;
  const bool masked = mask != nullptr;
  const int ring = f3_ring(L, sh, masked, lds_limit);
  if (ring < 4) return hipErrorInvalidValue;
  const F3Carve cv = f3_carve(L, sh, masked, ring);
  hipError_t e = hipErrorInvalidValue;
  if (ev0 && (e = hipEventRecord(ev0, stream)) != hipSuccess) return e;
#define GO(NBLK_, NBB_) e = f3_launch_m<NBLK_, NBB_>(sh.me, num, cat, rows, L, D, cv, ring, grid, partials, pair_slabs, skip, mask, kept, stream)
  switch (10 * sh.nblk + sh.nbb) {                            // (nblk, nbb) pairs that exist for n = 1..20
#ifndef F3_DEV_ONLY_10_10
    case 11: GO(1, 1); break; case 21: GO(2, 1); break; case 22: GO(2, 2); break;
    case 33: GO(3, 3); break; case 43: GO(4, 3); break; case 44: GO(4, 4); break; case 54: GO(5, 4); break;
#endif
    case 32: GO(3, 2); break;
    default: break;
  }
#undef GO
  if (e != hipSuccess) return e;
  if (ev1 && (e = hipEventRecord(ev1, stream)) != hipSuccess) return e;
  if ((e = launch_pairs_fold2(pair_slabs, grid, L.n_p, D.p, stream)) != hipSuccess) return e;
  return launch_gram_fold(partials, grid, acc, stream);
}

}  // namespace cofactor
