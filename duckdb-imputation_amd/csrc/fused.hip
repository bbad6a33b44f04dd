// sum_to_triple_n_m with low-cardinality categorical columns (every column <= 16 distinct keys):
// ONE pass over the n float and m int32 columns produces the dense part (lin_agg, quad_agg) and
// all categorical tables (lin_cat, quad_num_cat, quad_cat).  Replaces, for these shapes, the
// gram_kernel + cat_accumulate pair (which reads the float columns twice and spends its time in
// 100 ds_add_f64 per row at 10_10).
//
// Reference loops replaced: duckdb_extension/src/triple/sum/sum_no_lift.cpp:119-214.
//
// Workgroup = 768 threads = THREE teams of 4 waves that meet at ONE barrier per 256-row tile, on tiles
// double-buffered in LDS, so loading, LDS-atomic work and matrix work of one CU overlap:
//   loaders   (waves 0-3) keep two tiles of 16-byte non-temporal loads in flight (register ring) and
//             park whole columns: float columns go to the Gram tile xt[col][row] and, split exactly
//             into three bf16 pieces (x = hi + mid + lo), to pt[piece * n + col][row]; key columns
//             are translated to codes (byte table in LDS for keys 0..255, else the LDS copy of the
//             column's hash dictionary) and go to codes[col][row] (u16);
//   counters  (waves 4-7), one thread per row: the m(m-1)/2 off-diagonal pair increments as
//             ds_add_u32 into pair tables in LDS (ONE u32 cell per key pair), flushed once at the end
//             into the workgroup's private slab in HBM; key counts and the diagonal pairs are row
//             sums of pair table (c, c+1) and are filled in by fused_pairs_fold_kernel (a single key
//             column counts in the row loop);
//   MFMA team (waves 8-11), 64 rows per wave: the dense Gram with v_mfma_f32_4x4x1_16b_f32 (as
//             gram.hip), and the per-key sums as ONE-HOT x PIECES products on
//             v_mfma_f32_32x32x16_bf16: A = one-hot(code) of two key columns (32 rows of A =
//             2 x 16 codes), B = the <= 32 piece columns, K = 16 table rows.  1.0 x piece is exact,
//             accumulation is fp32 and is folded into the fp64 LDS table every S_FLUSH_TILES = 32
//             tiles (the chains add bf16 pieces, 8-bit mantissas: all but exact, guarded by a test).
// At the end the sum tables are added to the aggregate's HBM tables with global atomics, the pair
// slabs are folded by fused_pairs_fold_kernel, and the Gram image goes through the same partials +
// gram_fold_kernel path as gram.hip.  (fused2.hip is the LDS-DMA / all-MFMA variant that also takes
// NB aggregates and n = 0; this kernel is the faster one on the shapes it takes.)
#include "device.hpp"

namespace cofactor {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int TR = GRAM_TILE_ROWS;      // 256 rows per tile, one thread per row
constexpr int XCS = GRAM_COL_STRIDE;    // float stride of an xt column
constexpr int PTS = TR + 8;             // u16 stride of a codes column (528 B, 16-B aligned)
constexpr int S_FLUSH_TILES = 32;        // tiles between fp64 folds of the per-key sums: a cell then holds at most
                                        // 2048 fp32 adds of bf16 pieces (8-bit mantissas, so mostly exact); measured
                                        // error with ONE key per column <= 4e-8 (tests/tools/s_error_probe.py)
constexpr int G_FLUSH_TILES = 4;        // x rows per MFMA: at most 64 fp32 adds per Gram chain between fp64 folds
constexpr int LOAD_RING = 2;            // tiles whose loads the loader team keeps in flight

__device__ __forceinline__ unsigned fhash(int32_t key, int cap) {
  return ((unsigned)key * 0x9E3779B1u) >> (32 - (31 - __builtin_clz(cap)));
}

// codes of 4 keys in the LDS copy of a dictionary (NO_CODE = 16 if absent), packed as 4 x u16.  The four
// probe chains advance together, so their LDS round trips overlap.  One copy in the code object
// (called 2.5 times per loader wave and tile): the loader loop has to stay small.
__device__ __forceinline__ uint2 lds_lookup4(const unsigned long long *slots, const int32_t *codes, int cap,
                                          uint4 keys) {
  const unsigned key[4] = {keys.x, keys.y, keys.z, keys.w};
  unsigned h[4], cd[4];
  bool open[4];
#pragma unroll
  for (int e = 0; e < 4; e++) { h[e] = fhash((int32_t)key[e], cap); open[e] = true; cd[e] = 16u; }   // NO_CODE
  for (int probe = 0; probe < cap; probe++) {
    unsigned long long cur[4];
#pragma unroll
    for (int e = 0; e < 4; e++) cur[e] = slots[h[e]];
    bool any = false;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const unsigned long long want = (1ull << 32) | (unsigned long long)key[e];
      if (open[e]) {
        if (cur[e] == want) { cd[e] = (unsigned)codes[h[e]] & 0xFFFFu; open[e] = false; }
        else if (cur[e] == 0ull) open[e] = false;
        else h[e] = (h[e] + 1) & (cap - 1);
      }
      any = any || open[e];
    }
    if (!any) break;
  }
  return make_uint2(cd[0] | (cd[1] << 16), cd[2] | (cd[3] << 16));
}

// 8 u16 codes (uint4) -> 8 bf16 one-hot values for code value i: 1.0 (0x3F80) where equal.
// Codes in the tile are 0..15, or NO_CODE = 16 for "no such key / no such column", and ii holds
// i (0..15) in both halves, so d = code ^ i is 0..31 in each 16-bit half and 0 only on a match:
// 0x20 - d has bit 5 set exactly there, with no borrow between the halves.  Plain 32-bit ops
// (hipcc scalarises 16-bit vector compares into v_cmp/v_cndmask/v_perm chains).
typedef __attribute__((address_space(3))) unsigned lds_u32;
constexpr int DIRECT_KEYS = 256;        // keys 0..255 of a column are looked up in a byte table
constexpr int DIRECT_STRIDE = 260;      // bytes per column: 256 keys + the NO_CODE entry every other key reads
constexpr unsigned NO_CODE = 16u;
constexpr unsigned ROW_OFF = 17u;   // code of every column of a row the row filter dropped (matches no i either)
__device__ __forceinline__ bf16x8 onehot8(uint4 cv, unsigned ii) {
  unsigned w[4] = {cv.x, cv.y, cv.z, cv.w};
  // per 16-bit half: d = code ^ i, min(d, 1) is 0 on a match and 1 otherwise, and
  // 0x3F80 + min * 0xC080 (mod 2^16) is bf16 1.0 or 0: three packed VALU ops per two rows.
  // (Inline asm: hipcc turns the same thing written with ushort2 vectors into scalar compares.)
  const unsigned ones = 0x00010001u, neg = 0xC080C080u, one_bf = 0x3F803F80u;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    unsigned m, r;
    const unsigned d = w[e] ^ ii;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(d), "v"(ones));
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(neg), "v"(one_bf));
    w[e] = r;
  }
  return __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
}

constexpr int FUSED_THREADS = 768;      // waves 0-3 loaders, 4-7 counters, 8-11 MFMA team
constexpr int TEAM = 256;

struct FusedCarve {   // byte offsets into the dynamic LDS block
  int xt, xt_stride, pt, pt_stride, codes, codes_stride, s, slot, dcode, cnt, pairs, nf, gsum, direct, total;
};

template <int NB, int NBB, int M>
__global__ __launch_bounds__(FUSED_THREADS) void fused_kernel(NumCols num, CatCols cat, uint64_t rows,
                                                              CatLayout L, CatDevice D, FusedCarve cv,
                                                              double *__restrict__ partials,
                                                              unsigned *__restrict__ pair_slabs,
                                                              unsigned *__restrict__ skip,
                                                              const uint8_t *__restrict__ mask,
                                                              unsigned long long *__restrict__ kept) {
  constexpr int NPAIR = NB * (NB + 1) / 2;
  constexpr int NBC = 4 * NB;
  constexpr int MP = (M + 1) / 2;                          // key columns are processed in pairs by the MFMA team
  constexpr int MC = M;                                    // number of key columns, compile-time
  constexpr int LDX = NB + (M + 3) / 4;                    // 16-B loads per loader thread per tile
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  double *l_s = reinterpret_cast<double *>(lds + cv.s);
  unsigned long long *l_slot = reinterpret_cast<unsigned long long *>(lds + cv.slot);
  int32_t *l_dcode = reinterpret_cast<int32_t *>(lds + cv.dcode);
  unsigned *l_cnt = reinterpret_cast<unsigned *>(lds + cv.cnt);
  unsigned *l_p = reinterpret_cast<unsigned *>(lds + cv.pairs);   // one u32 cell per key pair
  unsigned *l_nf = reinterpret_cast<unsigned *>(lds + cv.nf);     // [buffer]: stamp of a tile holding inf / nan
  unsigned *l_skip = l_nf + 2;                                    // [buffer]: stamp of a tile with an unknown key
  unsigned char *l_direct = lds + cv.direct;                      // [column][key 0..255] -> code (NO_CODE if absent)
  unsigned char *l_far = l_direct + M * DIRECT_STRIDE;              // [column]: the dictionary holds a key outside 0..255
  auto xt_of = [&](int b) { return reinterpret_cast<float *>(lds + cv.xt + b * cv.xt_stride); };
  auto pt_of = [&](int b) { return reinterpret_cast<unsigned short *>(lds + cv.pt + b * cv.pt_stride); };
  auto codes_of = [&](int b) { return reinterpret_cast<unsigned short *>(lds + cv.codes + b * cv.codes_stride); };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2;                              // 0 loaders, 1 counters, 2 MFMA
  const int tw = wave & 3;                                 // wave index inside its team
  const int tt = tid & (TEAM - 1);                         // thread index inside its team
  const int n = L.n;
  constexpr int m = M;
  const int n_pw = L.n_p;                                  // dwords of the pair tables (one u32 cell each)

  // ---- one-time LDS setup ------------------------------------------------------------------
  for (int b = 0; b < 2; b++) {
    float *x = xt_of(b);
    unsigned short *cd = codes_of(b);
    for (int i = tid; i < (NBC + 1) * XCS; i += FUSED_THREADS) x[i] = 0.f;
    unsigned short *pz = pt_of(b);
    for (int i = tid; i < 32 * NBB * PTS; i += FUSED_THREADS) pz[i] = 0;   // piece columns >= 3n stay 0
    for (int i = tid; i < (m + 1) * PTS; i += FUSED_THREADS) cd[i] = (unsigned short)NO_CODE;   // column m stays NO_CODE
  }
  for (int i = tid; i < L.n_slots; i += FUSED_THREADS) { l_slot[i] = D.ht_slot[i]; l_dcode[i] = D.ht_code[i]; }
  for (int i = tid; i < L.n_cnt; i += FUSED_THREADS) l_cnt[i] = 0u;
  for (int i = tid; i < L.n_s; i += FUSED_THREADS) l_s[i] = 0.0;
  for (int i = tid; i < n_pw; i += FUSED_THREADS) l_p[i] = 0u;
  if (tid < 4) l_nf[tid] = 0u;                               // l_nf[0..1], l_skip[0..1]
  // Small non-negative keys (the usual encoding of a categorical column) skip the hash probe: a
  // byte table per column maps key 0..255 to its code.  A column whose dictionary holds any
  // other key keeps probing (flag per column).
  for (int i = tid; i < M * DIRECT_STRIDE; i += FUSED_THREADS) l_direct[i] = (unsigned char)NO_CODE;
  if (tid < 32) l_far[tid] = 0;
  __syncthreads();
  for (int c = 0; c < m; c++)
    for (int i = tid; i < L.ht_cap[c]; i += FUSED_THREADS) {
      const unsigned long long sv = l_slot[L.ht_off[c] + i];
      const int32_t cdv = l_dcode[L.ht_off[c] + i];
      if (sv != 0ull && cdv >= 0) {
        const unsigned key = (unsigned)(sv & 0xFFFFFFFFull);
        if (key < (unsigned)DIRECT_KEYS) l_direct[c * DIRECT_STRIDE + key] = (unsigned char)cdv;
        else l_far[c] = 1;
      }
    }

  // ---- lane roles of the MFMA team -------------------------------------------------------------
  // Gram operand columns as in gram.hip: block b serves block pair b % NPAIR of row group
  // b / NPAIR (RPM row groups per MFMA when the triangle leaves blocks of the instruction free)
  constexpr int RPM = NPAIR <= 1 ? 16 : (NPAIR <= 3 ? 4 : (NPAIR <= 6 ? 2 : 1));
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
  const int r32 = lane & 31, h32 = lane >> 5;              // 32x32x16 operand roles
  const unsigned ii = (unsigned)(r32 & 15) * 0x00010001u;  // this lane's code value, twice
  int s_k[NBB];                                            // numeric column of this lane's piece column
  bool s_ok[NBB];
#pragma unroll
  for (int bb = 0; bb < NBB; bb++) {
    const int pc = 32 * bb + r32;                          // piece column = piece * n + numeric column
    s_ok[bb] = pc < 3 * n;
    s_k[bb] = s_ok[bb] ? pc % n : 0;
  }

#ifdef COFACTOR_DEV_ABLATE   // timing experiments only (results are wrong when a bit is set)
  const int ablate = D.flags[2];      // 1: no phase 2, 2: no phase 3b, 4: no Gram, 8: no lookups
#else
  constexpr int ablate = 0;
#endif
  const uint64_t ntiles = rows / TR;                        // whole tiles only

  // ---- loader team: tile fetch (global -> registers) and park (registers -> LDS) ---------------
  // (the launcher only hands this kernel whole tiles of 16-byte aligned columns)
  auto fetch = [&](uint4 (&pre)[LDX], unsigned &pre_mask, uint64_t t) {
    const uint64_t r0 = t * TR + 4 * (uint64_t)lane;
    // row filter of this lane's 4 rows, one byte each (the launcher checked the 4-byte alignment)
    // (without a filter the same instruction re-reads the first 256 bytes of column 0 — always the
    // same cache line, no HBM traffic — and the word is ignored: the number of loads per call stays
    // fixed, see below)
    const uint8_t *mbase = mask ? mask + r0 : reinterpret_cast<const uint8_t *>(n ? (const void *)num.p[0] : (const void *)cat.p[0]) + 4 * lane;
    // raw word: park() decides whether it means anything (a select here would make the wave wait
    // for this load, and so for every older load of the ring, right after issuing it)
    pre_mask = *reinterpret_cast<const unsigned *>(mbase);
    // Every slot loads, a slot past the last column re-reads the last one: a fixed number of
    // loads per call lets the compiler wait with vmcnt(N) for the tile it parks and leave the
    // younger tile's loads in flight (a conditional load forces vmcnt(0), i.e. a ring of one).
#pragma unroll
    for (int i = 0; i < LDX; i++) {
      const int vc = min(tw + 4 * i, n + m - 1);           // wave-uniform virtual column
      const unsigned *src = vc < n ? reinterpret_cast<const unsigned *>(num.p[vc])
                                   : reinterpret_cast<const unsigned *>(cat.p[vc - n]);
      const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src + r0));
      pre[i] = __builtin_bit_cast(uint4, v);
    }
  };
  auto park = [&](const uint4 (&pre)[LDX], unsigned raw_mask, int b, unsigned stamp) {
    const unsigned pre_mask = mask ? raw_mask : 0x01010101u;   // row filter of this lane's 4 rows
    float *xt = xt_of(b);
    unsigned short *pt = pt_of(b);
    unsigned short *codes = codes_of(b);
#pragma unroll
    for (int i = 0; i < LDX; i++) {
      const int vc = tw + 4 * i;
      if (vc < n) {
        // filtered rows contribute x = 0 to the Gram and to the per-key sums
        unsigned u[4] = {pre[i].x, pre[i].y, pre[i].z, pre[i].w};
        if (mask) {
          u[0] = (pre_mask & 0x000000FFu) ? u[0] : 0u; u[1] = (pre_mask & 0x0000FF00u) ? u[1] : 0u;
          u[2] = (pre_mask & 0x00FF0000u) ? u[2] : 0u; u[3] = (pre_mask & 0xFF000000u) ? u[3] : 0u;
        }
        *reinterpret_cast<uint4 *>(&xt[vc * XCS + 4 * lane]) = make_uint4(u[0], u[1], u[2], u[3]);
        // x = hi + mid + lo, each a bf16 (exact): hi = upper half of x, mid = upper half of
        // x - hi, lo = x - hi - mid (its lower half is zero).  v_perm_b32 packs the upper halves
        // of two values into one dword.
        constexpr unsigned UPPER_HALVES = 0x07060302u;      // {s0.b3, s0.b2, s1.b3, s1.b2}
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
        // inf / nan (rare): their pieces become 0 here and the value is added to its own key's
        // cells by the counters (0 x inf would poison every key's cell)
        constexpr int INF_NAN = 0x207;                      // sNaN | qNaN | -inf | +inf
        const bool bad0 = __builtin_amdgcn_classf(__uint_as_float(u[0]), INF_NAN), bad1 = __builtin_amdgcn_classf(__uint_as_float(u[1]), INF_NAN);
        const bool bad2 = __builtin_amdgcn_classf(__uint_as_float(u[2]), INF_NAN), bad3 = __builtin_amdgcn_classf(__uint_as_float(u[3]), INF_NAN);
        const bool nonfinite = bad0 || bad1 || bad2 || bad3;
        if (nonfinite) {
          const unsigned k0 = (bad0 ? 0u : 0x0000FFFFu) | (bad1 ? 0u : 0xFFFF0000u);
          const unsigned k1 = (bad2 ? 0u : 0x0000FFFFu) | (bad3 ? 0u : 0xFFFF0000u);
          ph.x &= k0; pm.x &= k0; pl.x &= k0;
          ph.y &= k1; pm.y &= k1; pl.y &= k1;
        }
        *reinterpret_cast<uint2 *>(&pt[vc * PTS + 4 * lane]) = ph;
        *reinterpret_cast<uint2 *>(&pt[(n + vc) * PTS + 4 * lane]) = pm;
        *reinterpret_cast<uint2 *>(&pt[(2 * n + vc) * PTS + 4 * lane]) = pl;
        if (nonfinite) l_nf[b] = stamp;                     // this tile holds inf / nan
      } else if (vc < n + m) {
        const int c = vc - n;
        const unsigned long long *slots = l_slot + L.ht_off[c];
        const int32_t *dc = l_dcode + L.ht_off[c];
        const int cap = L.ht_cap[c];
        uint2 packed;
        if (ablate & 8) packed = make_uint2((pre[i].x & 15u) | ((pre[i].y & 15u) << 16), (pre[i].z & 15u) | ((pre[i].w & 15u) << 16));
        else if (!l_far[c]) {                               // wave-uniform: byte table
          const unsigned char *dt = l_direct + c * DIRECT_STRIDE;
          // any key outside 0..255 (negative ones are huge as unsigned) reads entry 256 = NO_CODE
          const unsigned cx = dt[min(pre[i].x, (unsigned)DIRECT_KEYS)], cy = dt[min(pre[i].y, (unsigned)DIRECT_KEYS)];
          const unsigned cz = dt[min(pre[i].z, (unsigned)DIRECT_KEYS)], cw = dt[min(pre[i].w, (unsigned)DIRECT_KEYS)];
          packed = make_uint2(cx | (cy << 16), cz | (cw << 16));
        } else packed = lds_lookup4(slots, dc, cap, pre[i]);
        *reinterpret_cast<uint2 *>(&codes[c * PTS + 4 * lane]) = packed;
        // optimistic mode (skip != nullptr): a tile that meets a key the dictionary does not
        // know yet is left out as a whole and redone by the host after a dictionary pass
        if (skip && ((packed.x | packed.y) & 0x00100010u))  // some code is NO_CODE
          l_skip[b] = stamp;
        if (mask) {                                         // filtered rows: code ROW_OFF in every column
          if (!(pre_mask & 0x000000FFu)) packed.x = (packed.x & 0xFFFF0000u) | ROW_OFF;
          if (!(pre_mask & 0x0000FF00u)) packed.x = (packed.x & 0x0000FFFFu) | (ROW_OFF << 16);
          if (!(pre_mask & 0x00FF0000u)) packed.y = (packed.y & 0xFFFF0000u) | ROW_OFF;
          if (!(pre_mask & 0xFF000000u)) packed.y = (packed.y & 0x0000FFFFu) | (ROW_OFF << 16);
          *reinterpret_cast<uint2 *>(&codes[c * PTS + 4 * lane]) = packed;
        }
      }
    }
  };
  // pair cells -> this workgroup's private u32 slab in HBM, once at the end (a u32 cell holds the
  // rows of any launch).  atomicExch makes read-and-clear safe against increments that run ahead.
  unsigned *slab = pair_slabs + (uint64_t)blockIdx.x * (uint64_t)n_pw;
  auto flush_pairs = [&]() {
    for (int w = tt; w < n_pw; w += TEAM) slab[w] = atomicExch(&l_p[w], 0u);
  };
  // counts and pair counts of tile t (buffer b), one loader thread per row
  unsigned n_kept = 0;                                     // rows this counter thread counted (masked updates)
  auto count_rows = [&](int b, unsigned stamp) {
    if (ablate & 1) return;
    const unsigned short *codes = codes_of(b);
    unsigned cd[MC];
    bool known = true;
#pragma unroll
    for (int c = 0; c < MC; c++)
      if (c < m) { cd[c] = codes[c * PTS + tt]; known = known && cd[c] < (unsigned)L.kc[c]; }
    if (cd[0] == ROW_OFF) return;                           // row filtered out by the mask
    n_kept++;
    if (!known) { D.flags[1] = 1; return; }                 // surfaces as an error at finalize
    // every column has code capacity 16 here (fused_applicable), so pair table q starts at cell
    // 256 q and cell = 256 q + 16 code1 + code2, one dword each (16-bit cells packed two per
    // dword halve the table but double the lanes that collide on a dword).  Diagonal pairs
    // (c, c) only ever hit cells (k, k) with the column's own counts: the fold kernel fills them.
    const unsigned lp_base = (unsigned)(unsigned long long)(lds_u32 *)l_p;   // 32-bit LDS address
    unsigned row8[MC], half[MC];
#pragma unroll
    for (int c = 0; c < MC; c++)
      if (c < m) {
        // counts: with two or more key columns they are row sums of a pair table and are taken
        // from there by fused_pairs_fold_kernel (16 cells per column make these the most
        // contended LDS atomics of the kernel); a single key column counts here
        if (M == 1) atomicAdd(&l_cnt[16 * c + cd[c]], 1u);
        row8[c] = 64u * cd[c];                              // byte offset of the row in a pair table
        half[c] = lp_base + 4u * cd[c];                     // LDS address of the cell inside row 0
        // keep the two in registers: left alone the compiler recomputes them from the code for
        // every one of the m(m-1)/2 pairs
        asm volatile("" : "+v"(row8[c]), "+v"(half[c]));
      }
    int q = 0;
#pragma unroll
    for (int c1 = 0; c1 < MC; c1++)
#pragma unroll
      for (int c2 = c1; c2 < MC; c2++)
        if (c2 < m) {
          if (c2 != c1)                                     // one v_add + ds_add_u32 offset:1024q
            __hip_atomic_fetch_add((lds_u32 *)(unsigned long long)(row8[c1] + half[c2] + 1024u * q),
                                   1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          q++;
        }
    if (l_nf[b] == stamp) {                                 // rare: pieces8 fed 0 for inf / nan
      const float *xt = xt_of(b);
      for (int k = 0; k < n; k++) {
        const float x = xt[k * XCS + tt];
        if ((__float_as_uint(x) & 0x7F800000u) == 0x7F800000u)
#pragma unroll
          for (int c = 0; c < MC; c++)
            if (c < m) unsafeAtomicAdd(&l_s[L.s_off[c] + (int)cd[c] * n + k], (double)x);
      }
    }
  };

  // ---- MFMA team state ---------------------------------------------------------------------------
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  f32x2 ls_lo = {0.f, 0.f}, ls_hi = {0.f, 0.f};
  double *gsum = reinterpret_cast<double *>(lds + cv.gsum) + tw * GRAM_ACC_LEN;   // this wave's fp64 Gram image
  f32x16 sacc[MP][NBB];
#pragma unroll
  for (int p = 0; p < MP; p++)
#pragma unroll
    for (int bb = 0; bb < NBB; bb++)
#pragma unroll
      for (int g = 0; g < 16; g++) sacc[p][bb][g] = 0.f;

  auto flush_gram = [&]() {                                // fp32 chains -> this wave's fp64 image in LDS
    gsum[0 * 64 + lane] += (double)((acc0[0] + acc1[0]) + (acc2[0] + acc3[0]));
    gsum[1 * 64 + lane] += (double)((acc0[1] + acc1[1]) + (acc2[1] + acc3[1]));
    gsum[2 * 64 + lane] += (double)((acc0[2] + acc1[2]) + (acc2[2] + acc3[2]));
    gsum[3 * 64 + lane] += (double)((acc0[3] + acc1[3]) + (acc2[3] + acc3[3]));
    gsum[4 * 64 + lane] += (double)((ls_lo[0] + ls_lo[1]) + (ls_hi[0] + ls_hi[1]));
    acc0 = acc1 = acc2 = acc3 = f32x4{0, 0, 0, 0};
    ls_lo = ls_hi = f32x2{0.f, 0.f};
  };
  auto flush_s = [&]() {
    // D register g of this lane is table cell (A-row i, B-column r32) with
    // i = (g&3) + 8*(g>>2) + 4*h32: key column 2p + (g>>3), code i & 15; B-column r32 is piece
    // column 32*bb + r32, i.e. numeric column s_k[bb].  Address = uniform part + one lane term.
#pragma unroll
    for (int bb = 0; bb < NBB; bb++) {
      const int lane_term = 4 * h32 * n + s_k[bb];
#pragma unroll
      for (int p = 0; p < MP; p++)
#pragma unroll
        for (int g = 0; g < 16; g++) {
          const int c = 2 * p + (g >> 3);
          const float v = sacc[p][bb][g];
          if (c < m && s_ok[bb] && v != 0.f) {
            const int uni = __builtin_amdgcn_readfirstlane(L.s_off[c] + ((g & 3) + 8 * ((g >> 2) & 1)) * n);
            unsafeAtomicAdd(&l_s[uni + lane_term], (double)v);
          }
          sacc[p][bb][g] = 0.f;
        }
    }
  };
  // Gram and per-key sums of this wave's 64 rows of the tile in buffer b
  auto crunch = [&](int b) {
    const float *xt = xt_of(b);
    const unsigned short *pt = pt_of(b);
    const unsigned short *codes = codes_of(b);
    if (!(ablate & 4)) {
      const f32x4 *va = reinterpret_cast<const f32x4 *>(xt + colA * XCS + tw * 64 + 4 * rsub);
      const f32x4 *vb = reinterpret_cast<const f32x4 *>(xt + colB * XCS + tw * 64 + 4 * rsub);
#pragma unroll 4
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
    if (!(ablate & 2))
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int row8 = tw * 64 + g * 16 + 8 * h32;        // this lane's 8 table rows
        bf16x8 bop[NBB];
#pragma unroll
        for (int bb = 0; bb < NBB; bb++)
          bop[bb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(&pt[(32 * bb + r32) * PTS + row8]));
#pragma unroll
        for (int p = 0; p < MP; p++) {
          int c = 2 * p + (r32 >> 4);
          c = c < m ? c : m;                                // column m is all NO_CODE: matches nothing
          const uint4 cvv = *reinterpret_cast<const uint4 *>(&codes[c * PTS + row8]);
          const bf16x8 aop = onehot8(cvv, ii);
#pragma unroll
          for (int bb = 0; bb < NBB; bb++)
            sacc[p][bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aop, bop[bb], sacc[p][bb], 0, 0, 0);
        }
      }
  };

  // ---- the pipeline.  Three teams meet at ONE barrier per tile: while counters and the MFMA team
  // consume tile k from buffer b, the loaders park tile k+1 into buffer b^1 (its loads were issued
  // two tiles ahead) and issue the loads of tile k+3.  Each team runs its own loop, so none carries
  // the others' registers; all execute the same number of barriers.
  for (int i = tid; i < 4 * GRAM_ACC_LEN; i += FUSED_THREADS) reinterpret_cast<double *>(lds + cv.gsum)[i] = 0.0;
  __syncthreads();                                          // LDS setup visible
  const uint64_t G = gridDim.x;
  if (team == 0) {
    // register ring of RING tiles: tile j of this workgroup lives in set j % RING from the moment
    // its loads are issued (RING tiles ahead of its use) until it is parked
    constexpr int RING = LOAD_RING;
    uint4 pre[RING][LDX];
    unsigned pmask[RING];
    uint64_t t = blockIdx.x;
    unsigned k = 1;                                         // per-workgroup tile counter = nf stamp
    // (ntiles >= 1 and blockIdx.x < ntiles: the launcher sizes the grid that way)
#pragma unroll
    for (int r = 0; r < RING; r++) fetch(pre[r], pmask[r], min(t + r * G, ntiles - 1));
    if (t < ntiles) park(pre[0], pmask[0], 0, k);
    fetch(pre[0], pmask[0], min(t + RING * G, ntiles - 1));
    __syncthreads();
    int b = 0;
    while (t < ntiles) {
#pragma unroll
      for (int r = 0; r < RING; r++) {                      // consuming tile j (j % RING == r)
        const int nx = (r + 1) % RING;
        if (t + G < ntiles) park(pre[nx], pmask[nx], b ^ 1, k + 1);
        fetch(pre[nx], pmask[nx], min(t + (RING + 1) * G, ntiles - 1));   // past the end: a harmless re-read
        __syncthreads();                                    // buffer b free, buffer b^1 complete
        t += G; b ^= 1; k++;
        if (t >= ntiles) break;
      }
    }
  } else if (team == 1) {
    __syncthreads();
    int b = 0;
    unsigned k = 1;
    for (uint64_t t = blockIdx.x; t < ntiles; t += G) {
      if (l_skip[b] == k) {                                 // tile left out: remember it for the host
        if (tt == 0) skip[1 + atomicAdd(&skip[0], 1u)] = (unsigned)t;
      } else {
        count_rows(b, k);
      }
      __syncthreads();
      b ^= 1; k++;
    }
    flush_pairs();
    if (mask) {                                             // N of a masked update = rows kept
      unsigned long long kk = n_kept;
      for (int off = 32; off > 0; off >>= 1) kk += __shfl_down(kk, off, 64);
      if (lane == 0 && kk) atomicAdd(kept, kk);
    }
  } else {
    __syncthreads();
    int b = 0, since_g = 0, since_s = 0;
    unsigned k = 1;
    for (uint64_t t = blockIdx.x; t < ntiles; t += G, k++) {
      if (l_skip[b] != k) crunch(b);
      if (++since_g == G_FLUSH_TILES * RPM) { flush_gram(); since_g = 0; }
      if (++since_s == S_FLUSH_TILES) { flush_s(); since_s = 0; }
      __syncthreads();
      b ^= 1;
    }
    flush_gram();
    flush_s();
  }

  // ---- end: Gram image of the workgroup, count / sum tables into the aggregate's HBM tables ---
  __syncthreads();
  const double *red = reinterpret_cast<const double *>(lds + cv.gsum);
  for (int i = tid; i < GRAM_ACC_LEN; i += FUSED_THREADS) {   // 4 waves and RPM row groups, fixed order
    double v = 0;
    if ((i & 63) < 4 * NPAIR)
#pragma unroll
      for (int rs = 0; rs < RPM; rs++) {
        const int j = i + 4 * NPAIR * rs;
        v += ((red[j] + red[GRAM_ACC_LEN + j]) + red[2 * GRAM_ACC_LEN + j]) + red[3 * GRAM_ACC_LEN + j];
      }
    partials[(uint64_t)i * gridDim.x + blockIdx.x] = v;
  }
  for (int i = tid; M == 1 && i < L.n_cnt; i += FUSED_THREADS)
    if (l_cnt[i]) {
      atomicAdd(&D.cnt[i], (unsigned long long)l_cnt[i]);
      const int c = i >> 4, code = i & 15;                  // cnt_off[c] = 16 c in the fused layout
      const int qd = c * m - c * (c - 1) / 2;               // index of pair (c, c)
      atomicAdd(&D.p[L.p_off[qd] + code * 16 + code], (unsigned long long)l_cnt[i]);
    }
  for (int i = tid; i < L.n_s; i += FUSED_THREADS)
    if (l_s[i] != 0.0) unsafeAtomicAdd(&D.s[i], l_s[i]);
}

// D.p[cell] += sum over workgroups of slab[wg][cell]
// With m >= 2 key columns the kernel keeps no count table: a column's counts are the row sums of
// its pair table with the next column (column sums for the last column), added here to cnt and to
// the diagonal cells (k, k) of the column's own pair table.  Fused layout: table q = 256 cells
// [code1][code2], counts of column c at cnt[16 c ..].
__global__ __launch_bounds__(256) void fused_pairs_fold_kernel(const unsigned *__restrict__ slabs, int nwg,
                                                               int cells_padded, int n_p, int m,
                                                               unsigned long long *__restrict__ p,
                                                               unsigned long long *__restrict__ cnt) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= n_p) return;
  unsigned long long total = 0;
  for (int w = 0; w < nwg; w++) total += slabs[(uint64_t)w * cells_padded + cell];
  if (!total) return;
  p[cell] += total;
  if (m < 2) return;
  const int q = cell >> 8, k1 = (cell >> 4) & 15, k2 = cell & 15;
  int c1 = 0, rem = q;
  while (rem >= m - c1) { rem -= m - c1; c1++; }           // q = index of (c1, c1 + rem), c2 >= c1
  if (rem != 1) return;                                     // only the tables (c, c + 1) carry counts
  const int qd1 = c1 * m - c1 * (c1 - 1) / 2;              // index of pair (c1, c1)
  atomicAdd(&cnt[16 * c1 + k1], total);
  atomicAdd(&p[256 * qd1 + 17 * k1], total);
  if (c1 + 1 == m - 1) {                                    // the last column has no successor
    const int c2 = m - 1, qd2 = c2 * m - c2 * (c2 - 1) / 2;
    atomicAdd(&cnt[16 * c2 + k2], total);
    atomicAdd(&p[256 * qd2 + 17 * k2], total);
  }
}

// temp[col][i * 256 + j] = col[list[i] * 256 + j]: the tiles the optimistic pass left out, packed
__global__ __launch_bounds__(256) void gather_tiles_kernel(NumCols num, CatCols cat, int n, int m,
                                                           const unsigned *__restrict__ list, unsigned count,
                                                           unsigned *__restrict__ temp, uint64_t temp_stride,
                                                           const uint8_t *__restrict__ mask,
                                                           uint8_t *__restrict__ temp_mask) {
  for (unsigned i = blockIdx.x; i < count; i += gridDim.x) {
    const uint64_t src = (uint64_t)list[i] * TR + threadIdx.x;
    const uint64_t dst = (uint64_t)i * TR + threadIdx.x;
    if (mask) temp_mask[dst] = mask[src];
    for (int c = 0; c < n + m; c++) {
      const unsigned *col = c < n ? reinterpret_cast<const unsigned *>(num.p[c])
                                  : reinterpret_cast<const unsigned *>(cat.p[c - n]);
      temp[(uint64_t)c * temp_stride + dst] = col[src];
    }
  }
}

FusedCarve make_carve(const CatLayout &L, int nb) {
  const int nbb = (3 * L.n + 31) / 32;
  FusedCarve c{};
  size_t o = 0;
  auto take = [&](size_t bytes, size_t align) { o = (o + align - 1) / align * align; size_t at = o; o += bytes; return (int)at; };
  c.xt_stride = (int)((size_t)(4 * nb + 1) * XCS * sizeof(float));
  c.xt = take((size_t)2 * c.xt_stride, 16);
  c.pt_stride = (int)((size_t)32 * nbb * PTS * 2);
  c.pt = take((size_t)2 * c.pt_stride, 16);
  c.codes_stride = (int)((size_t)(L.m + 1) * PTS * 2);
  c.codes = take((size_t)2 * c.codes_stride, 16);
  c.s = take((size_t)L.n_s * 8, 8);
  c.slot = take((size_t)L.n_slots * 8, 8);
  c.dcode = take((size_t)L.n_slots * 4, 4);
  c.cnt = take((size_t)L.n_cnt * 4, 4);
  c.pairs = take((size_t)L.n_p * 4, 4);
  c.nf = take(16, 4);
  c.gsum = take(sizeof(double) * 4 * GRAM_ACC_LEN, 8);
  c.direct = take((size_t)L.m * DIRECT_STRIDE + 32, 4);     // per key column: code of key 0..255, then the flags
  c.total = (int)((o + 15) / 16 * 16);
  return c;
}

template <int NB, int NBB, int M>
hipError_t launch_one(const NumCols &num, const CatCols &cat, uint64_t rows, const CatLayout &L,
                      const CatDevice &D, const FusedCarve &cv, int grid, double *partials,
                      unsigned *slabs, unsigned *skip, const uint8_t *mask, unsigned long long *kept,
                      hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute((const void *)fused_kernel<NB, NBB, M>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, cv.total);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((fused_kernel<NB, NBB, M>), dim3(grid), dim3(FUSED_THREADS), cv.total, stream, num,
                     cat, rows, L, D, cv, partials, slabs, skip, mask, kept);
  return hipGetLastError();
}

template <int NB, int NBB>
hipError_t launch_m(int m, const NumCols &num, const CatCols &cat, uint64_t rows, const CatLayout &L,
                     const CatDevice &D, const FusedCarve &cv, int grid, double *partials,
                     unsigned *slabs, unsigned *skip, const uint8_t *mask, unsigned long long *kept,
                     hipStream_t stream) {
  switch (m) {
#define CASE(M_) case M_: if constexpr (((M_ + 1) / 2) * NBB <= FUSED_MAX_SBLOCKS) return launch_one<NB, NBB, M_>(num, cat, rows, L, D, cv, grid, partials, slabs, skip, mask, kept, stream); else break;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10)
#undef CASE
    default: break;
  }
  return hipErrorInvalidValue;
}

}  // namespace

bool fused_applicable(const CatLayout &L, const int32_t *nkeys, size_t lds_limit, size_t *lds_bytes) {
  if (L.kind != 0 || L.n < 1 || L.m < 1) return false;
  for (int c = 0; c < L.m; c++)
    if (nkeys[c] > 16 || L.kc[c] != 16) return false;
  const int nb = (L.n + 3) / 4, nbb = (3 * L.n + 31) / 32, mp = (L.m + 1) / 2;
  if (mp * nbb > FUSED_MAX_SBLOCKS || L.m > 10) return false;
  const FusedCarve cv = make_carve(L, nb);
  if (lds_bytes) *lds_bytes = (size_t)cv.total;
  return (size_t)cv.total <= lds_limit;
}

int fused_grid(const CatLayout &L, int cus, int partials_cap_wgs, uint64_t rows) {
  int grid = cus;                                           // 12 waves of up to 168 registers and ~159 KB of LDS fill a CU
  if (grid > partials_cap_wgs) grid = partials_cap_wgs;
  const uint64_t ntiles = rows / TR;
  if ((uint64_t)grid > ntiles) grid = (int)ntiles;
  return grid;
}

size_t fused_slab_bytes(const CatLayout &L, int grid) {
  return (size_t)grid * (size_t)L.n_p * sizeof(unsigned);
}

hipError_t launch_fused(const NumCols &num, const CatCols &cat, uint64_t rows, const CatLayout &L,
                        const CatDevice &D, int grid, double *partials, unsigned *pair_slabs,
                        unsigned *skip, double *acc, hipStream_t stream, hipEvent_t ev0,
                        hipEvent_t ev1, const uint8_t *mask, unsigned long long *kept) {
  if (rows == 0) return hipSuccess;
  const int nb = (L.n + 3) / 4, nbb = (3 * L.n + 31) / 32;
  const FusedCarve cv = make_carve(L, nb);
  hipError_t e = hipErrorInvalidValue;
  if (ev0 && (e = hipEventRecord(ev0, stream)) != hipSuccess) return e;
#define GO(NB_, NBB_) e = launch_m<NB_, NBB_>(L.m, num, cat, rows, L, D, cv, grid, partials, pair_slabs, skip, mask, kept, stream)
  if (nbb == 1) {
    switch (nb) { case 1: GO(1, 1); break; case 2: GO(2, 1); break; case 3: GO(3, 1); break; default: break; }
  } else {
    switch (nb) { case 3: GO(3, 2); break; case 4: GO(4, 2); break; case 5: GO(5, 2); break; default: break; }
  }
#undef GO
  if (e != hipSuccess) return e;
  if (ev1 && (e = hipEventRecord(ev1, stream)) != hipSuccess) return e;
  const int cells_padded = L.n_p;
  hipLaunchKernelGGL(fused_pairs_fold_kernel, dim3((L.n_p + 255) / 256), dim3(256), 0, stream, pair_slabs,
                     grid, cells_padded, L.n_p, L.m, D.p, D.cnt);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return launch_gram_fold(partials, grid, acc, stream);
}

hipError_t launch_gather_tiles(const NumCols &num, const CatCols &cat, int n, int m, const unsigned *list,
                               unsigned count, unsigned *temp, uint64_t temp_stride, hipStream_t stream,
                               const uint8_t *mask, uint8_t *temp_mask) {
  if (count == 0) return hipSuccess;
  hipLaunchKernelGGL(gather_tiles_kernel, dim3(count < 4096 ? count : 4096), dim3(256), 0, stream, num, cat,
                     n, m, list, count, temp, temp_stride, mask, temp_mask);
  return hipGetLastError();
}

}  // namespace cofactor
