// sum_to_nb_agg_n_m (Triple::sum_to_nb_agg, duckdb_extension/src/triple/sum/sum_to_nb_agg.cpp:39-146)
// in one pass on the LDS-DMA ring: N, lin_agg, the DIAGONAL of quad_agg (:107-117) and the per-key
// counts of every key column (:124-145).
//
// The Naive-Bayes aggregate needs no matrix product at all: per row n column sums, n squares and m
// counter increments.  fused2_kernel's MODE 0 computes it with the machinery of the triple kind
// (Gram on MFMA, key counts as a one-hot product: 2.2-2.3 ms per 1e8 rows at 10_10, 44-46 % of the
// HBM peak).  Here (round 3) one 512-thread workgroup per CU has
//   waves 0-3: loaders — wait (counted vmcnt), barrier, issue the LDS-DMA of the tile `ring - 1`
//              ahead (SGPR base + one lane-offset VGPR), nothing else;
//   waves 4-7: one ROW PER LANE of their 64 rows of the tile: keys -> codes through the byte table
//              in LDS (hash probe for keys outside 0..255), one ds_add_u32 per key column into
//              the workgroup's count tables, and x, x^2 into fp32 lane sums (folded into fp64
//              every 64 tiles) — about 110 instructions per 64 rows, so the kernel runs at the
//              ring's pace.
// Any cardinality whose count tables and dictionaries fit LDS next to the ring (sum of the code
// capacities <= 12 K).  Row filter and optimistic mode (unknown key -> the 64-row block is left out
// and listed) as in the other one-pass kernels.
#include <cstdio>
#include <cstdlib>

#include "onepass.hpp"

namespace cofactor {
using namespace onepass;

namespace {

constexpr int NBR_THREADS = 512;
constexpr int NBR_FLUSH_TILES = 64;        // fp32 lane chains: one add per tile -> 64 terms between fp64 folds

struct NbrCarve { int ring, slot_bytes, cnt, direct, slot, dcode, total; };

// N4 / M4: n and m rounded up to multiples of 4 (compile-time loop bounds: every LDS read of a phase is
// issued before its first use — with run-time bounds each column was its own basic block and its two
// dependent LDS round trips were exposed: 5.7 ms instead of 1.5; columns past the last one re-read it)
template <int N4, int M4>
__global__ __launch_bounds__(NBR_THREADS) void nb_ring_kernel(NumCols num, CatCols cat, uint64_t rows, CatLayout L, CatDevice D,
                                                              NbrCarve cv, int ring, double *__restrict__ partials,
                                                              unsigned *__restrict__ skip, const uint8_t *__restrict__ mask,
                                                              unsigned long long *__restrict__ kept) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave < 4;
  const int sw = wave & 3;
  const int n = L.n, m = L.m;
  const bool masked = mask != nullptr;
  const int ndata = n + m, ncols = ndata + (masked ? 1 : 0);
  unsigned *l_cnt = reinterpret_cast<unsigned *>(lds + cv.cnt);
  unsigned char *l_direct = lds + cv.direct;
  unsigned char *l_far = l_direct + m * DIRECT_STRIDE * 2;
  unsigned short *l_direct16 = reinterpret_cast<unsigned short *>(l_direct);   // u16 codes: capacities past 255
  unsigned *l_kc = reinterpret_cast<unsigned *>(l_far + 32);                   // per column: code capacity, start of its count table
  unsigned *l_coff = l_kc + COFACTOR_MAX_CAT;
  unsigned long long *l_slot = reinterpret_cast<unsigned long long *>(lds + cv.slot);
  int32_t *l_dcode = reinterpret_cast<int32_t *>(lds + cv.dcode);
  constexpr unsigned NONE16 = 0xFFFFu;

  // ---- one-time LDS setup ----
  for (int i = tid; i < (cv.total - cv.cnt) / 4; i += NBR_THREADS) reinterpret_cast<unsigned *>(lds + cv.cnt)[i] = 0u;
  __syncthreads();
  for (int i = tid; i < L.n_slots; i += NBR_THREADS) { l_slot[i] = D.ht_slot[i]; l_dcode[i] = D.ht_code[i]; }
  for (int i = tid; i < m * DIRECT_STRIDE; i += NBR_THREADS) l_direct16[i] = (unsigned short)NONE16;
  if (tid < COFACTOR_MAX_CAT) { const int cc = min(tid, m - 1); l_kc[tid] = (unsigned)L.kc[cc]; l_coff[tid] = (unsigned)L.cnt_off[cc]; }
  __syncthreads();
  for (int c = 0; c < m; c++)
    for (int i = tid; i < L.ht_cap[c]; i += NBR_THREADS) {
      const unsigned long long sv = l_slot[L.ht_off[c] + i];
      const int32_t cdv = l_dcode[L.ht_off[c] + i];
      if (sv != 0ull && cdv >= 0) {
        const unsigned key = (unsigned)(sv & 0xFFFFFFFFull);
        if (key < (unsigned)DIRECT_KEYS) l_direct16[c * DIRECT_STRIDE + key] = (unsigned short)cdv;
        else l_far[c] = 1;
      }
    }
  __syncthreads();

  const uint64_t ntiles = rows / TR, G = gridDim.x;
  if (loader) {
    constexpr int MAXCPW = (COFACTOR_MAX_NUM + COFACTOR_MAX_CAT + 1 + 3) / 4;
    const int cpw = (ncols - sw + 3) / 4;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_void *)lds;
    const unsigned lane16 = 16u * lane, lane4 = 4u * lane;
    auto src_of = [&](int vc) -> const unsigned char * {
      return vc < n ? reinterpret_cast<const unsigned char *>(num.p[min(vc, COFACTOR_MAX_NUM - 1)])
                    : (vc < ndata ? reinterpret_cast<const unsigned char *>(cat.p[min(max(vc - n, 0), COFACTOR_MAX_CAT - 1)])
                                  : reinterpret_cast<const unsigned char *>(mask));
    };
    auto dma_tile = [&](uint64_t t, int slot) {
      const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + cv.ring + slot * cv.slot_bytes);
#pragma unroll 1
      for (int i = 0; i < MAXCPW; i++) {
        if (i >= cpw) break;
        const int vc = sw + 4 * i;
        const unsigned dst = __builtin_amdgcn_readfirstlane(base + vc * COLB);
        if (vc < ndata) glds16_s(src_of(vc) + t * (TR * 4), lane16, dst);
        else glds4_s(src_of(vc) + t * TR, lane4, dst);
      }
    };
    uint64_t t = blockIdx.x;
    for (int r = 0; r < ring - 1; r++) dma_tile(min(t + r * G, ntiles - 1), r);
    int slot = 0;
    const int keep = (ring - 2) * cpw;
    for (; t < ntiles; t += G) {
      wait_vmcnt(keep);
      __builtin_amdgcn_s_barrier();
      int nslot = slot + ring - 1;
      nslot = nslot >= ring ? nslot - ring : nslot;
      dma_tile(min(t + (uint64_t)(ring - 1) * G, ntiles - 1), nslot);
      slot = slot + 1 == ring ? 0 : slot + 1;
    }
    wait_vmcnt_imm<0>();
    __syncthreads();
    __syncthreads();
  } else {
    constexpr int NA = N4 > 0 ? N4 : 1;
    float fs[NA], fq[NA];
    double ds[NA], dq[NA];
#pragma unroll
    for (int k = 0; k < NA; k++) { fs[k] = fq[k] = 0.f; ds[k] = dq[k] = 0.0; }
    // per column (clamped to the last real one) its code capacity and the start of its count table, in
    // registers: read from the kernel arguments inside the loop they cost one scalar load + wait per
    // column and tile (the wait also drains the LDS reads in flight: 3.2 ms instead of 1.6)
    // (through LDS: straight from the kernel arguments hipcc keeps them in SGPRs, spills them and
    // re-loads them per tile all the same)
    unsigned kcv[M4], coff[M4];
#pragma unroll
    for (int c = 0; c < M4; c++) { kcv[c] = l_kc[c]; coff[c] = l_coff[c]; }
    unsigned farmask = 0;                                    // bit c: column c holds keys outside 0..255
    for (int c = 0; c < M4; c++) farmask |= (l_far[min(c, m - 1)] ? 1u : 0u) << c;
    farmask = __builtin_amdgcn_readfirstlane(farmask);
    unsigned n_kept = 0;
    int slot = 0, since = 0;
    for (uint64_t t = blockIdx.x; t < ntiles; t += G) {
      __builtin_amdgcn_s_barrier();                          // tile t is in `slot`
      const unsigned char *base = lds + cv.ring + slot * cv.slot_bytes;
      const int row = sw * 64 + lane;
      // ---- every read of the tile first: filter byte, keys, values ----
      unsigned fbyte = 1u;
      if (masked) fbyte = base[ndata * COLB + row];
      unsigned key[M4];
#pragma unroll
      for (int c = 0; c < M4; c++) key[c] = *reinterpret_cast<const unsigned *>(base + (n + min(c, m - 1)) * COLB + row * 4);
      float x[NA];
#pragma unroll
      for (int k = 0; k < N4; k++) x[k] = *reinterpret_cast<const float *>(base + min(k, n - 1) * COLB + row * 4);
      // ---- keys -> codes: byte-table lookups of all columns together, the hash probe only where needed ----
      unsigned cd[M4];
#pragma unroll
      for (int c = 0; c < M4; c++)
        cd[c] = l_direct16[min(c, m - 1) * DIRECT_STRIDE + min(key[c], (unsigned)DIRECT_KEYS)];
      bool keep_row = fbyte != 0;
      // (bit arithmetic, not && / ||: the short-circuit forms became twelve nested branches)
      unsigned probe = 0;
#pragma unroll
      for (int c = 0; c < M4; c++) probe |= (unsigned)(cd[c] == NONE16) & ((farmask >> c) & 1u);
      if (farmask != 0 && __builtin_amdgcn_ballot_w64(probe != 0) != 0ull) {
#pragma unroll
        for (int c = 0; c < M4; c++) {
          if (!((cd[c] == NONE16) & (((farmask >> c) & 1u) != 0))) continue;
          const int cc = min(c, m - 1);
          const unsigned long long want = (1ull << 32) | (unsigned long long)key[c];
          const unsigned long long *sl = l_slot + L.ht_off[cc];
          const int cap = L.ht_cap[cc];
          unsigned h = fhash2((int32_t)key[c], cap);
          for (int pr = 0; pr < cap; pr++) {
            const unsigned long long cur = sl[h];
            if (cur == want) { const int32_t v = l_dcode[L.ht_off[cc] + h]; cd[c] = v >= 0 ? (unsigned)v : NONE16; break; }
            if (cur == 0ull) break;
            h = (h + 1) & (cap - 1);
          }
        }
      }
      unsigned unknown = 0;
#pragma unroll
      for (int c = 0; c < M4; c++) {
        cd[c] = cd[c] >= kcv[c] ? NONE16 : cd[c];
        unknown |= (unsigned)(cd[c] == NONE16);
      }
      if (__builtin_amdgcn_ballot_w64((unknown != 0) & keep_row) != 0ull) {   // the whole 64-row block is left out and redone by the host
        if (lane == 0) {
          if (skip) skip[1 + atomicAdd(&skip[0], 1u)] = (unsigned)(t * 4 + sw);
          else D.flags[1] = 1;
        }
        keep_row = false;
      }
      if (keep_row) {
        n_kept++;
#pragma unroll
        for (int c = 0; c < M4; c++) atomicAdd(&l_cnt[coff[c] + cd[c]], c < m ? 1u : 0u);   // (a column past the last adds 0)
      }
#pragma unroll
      for (int k = 0; k < N4; k++) {
        const float v = keep_row ? x[k] : 0.f;
        fs[k] += v;
        fq[k] += v * v;
      }
      if (++since == NBR_FLUSH_TILES) {
#pragma unroll
        for (int k = 0; k < N4; k++) { ds[k] += (double)fs[k]; dq[k] += (double)fq[k]; fs[k] = fq[k] = 0.f; }
        since = 0;
      }
      slot = slot + 1 == ring ? 0 : slot + 1;
    }
#pragma unroll
    for (int k = 0; k < N4; k++) { ds[k] += (double)fs[k]; dq[k] += (double)fq[k]; }
    __syncthreads();                                         // (the loaders' drain: the ring is free)
    // wave sums in a fixed order, then the 4 compute waves' images into the ring area
    double *img = reinterpret_cast<double *>(lds + cv.ring) + sw * GRAM_ACC_LEN;
    for (int i = lane; i < GRAM_ACC_LEN; i += 64) img[i] = 0.0;
#pragma unroll
    for (int k = 0; k < N4; k++)
      if (k < n) {                                           // (columns past the last one re-read it: not stored)
        double a = ds[k], b = dq[k];
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
        if (lane == 0) { img[gram_lin_pos(k, n)] = a; img[gram_quad_pos(k, k, n)] = b; }
      }
    unsigned long long kk = n_kept;
    for (int off = 32; off > 0; off >>= 1) kk += __shfl_down(kk, off, 64);
    if (lane == 0 && kk && kept && masked) atomicAdd(kept, kk);
    __syncthreads();
  }
  const double *red = reinterpret_cast<const double *>(lds + cv.ring);
  for (int i = tid; i < GRAM_ACC_LEN; i += NBR_THREADS)
    partials[(uint64_t)i * gridDim.x + blockIdx.x] = ((red[i] + red[GRAM_ACC_LEN + i]) + red[2 * GRAM_ACC_LEN + i]) + red[3 * GRAM_ACC_LEN + i];
  for (int i = tid; i < L.n_cnt; i += NBR_THREADS)
    if (l_cnt[i]) atomicAdd(&D.cnt[i], (unsigned long long)l_cnt[i]);
}

bool nbr_carve(const CatLayout &L, bool masked, size_t lds_limit, NbrCarve &c, int &ring) {
  if (L.kind != COFACTOR_NB || L.m < 1 || L.n < 0) return false;
  if (L.n_cnt > 12 * 1024 || L.n_slots > 4096) return false;
  for (int cidx = 0; cidx < L.m; cidx++)
    if (L.kc[cidx] > 0xFFF0) return false;
  const int slot_bytes = ((L.n + L.m) * COLB + (masked ? 256 : 0) + 15) / 16 * 16;
  for (int r = 8; r >= 3; r--) {
    size_t o = 0;
    auto take = [&](size_t bytes, size_t align) { o = (o + align - 1) / align * align; size_t at = o; o += bytes; return (int)at; };
    NbrCarve t{};
    t.slot_bytes = slot_bytes;
    const size_t ring_bytes = std::max((size_t)r * slot_bytes, sizeof(double) * 4 * GRAM_ACC_LEN);
    t.ring = take(ring_bytes, 16);
    t.cnt = take((size_t)std::max(1, L.n_cnt) * 4, 4);          // (zeroed from here on)
    t.direct = take((size_t)L.m * DIRECT_STRIDE * 2 + 32 + 2 * COFACTOR_MAX_CAT * 4, 4);
    t.slot = take((size_t)L.n_slots * 8, 8);
    t.dcode = take((size_t)L.n_slots * 4, 4);
    t.total = (int)((o + 15) / 16 * 16);
    const int cpw = (L.n + L.m + (masked ? 1 : 0) + 3) / 4;
    if ((size_t)t.total <= lds_limit && (r - 2) * cpw <= 48) { c = t; ring = r; return true; }
  }
  return false;
}

}  // namespace

bool nbring_applicable(const CatLayout &L, bool masked, size_t lds_limit) {
  NbrCarve c;
  int ring;
  return nbr_carve(L, masked, lds_limit, c, ring);
}

hipError_t launch_nbring(const NumCols &num, const CatCols &cat, uint64_t rows, const CatLayout &L, const CatDevice &D,
                         int grid, size_t lds_limit, double *partials, unsigned *skip, double *acc, hipStream_t stream,
                         hipEvent_t ev0, hipEvent_t ev1, const uint8_t *mask, unsigned long long *kept) {
  if (rows == 0) return hipSuccess;
  NbrCarve cv;
  int ring;
  if (!nbr_carve(L, mask != nullptr, lds_limit, cv, ring)) return hipErrorInvalidValue;
  hipError_t e;
  if (ev0 && (e = hipEventRecord(ev0, stream)) != hipSuccess) return e;
  e = hipErrorInvalidValue;
#define GO(N4_, M4_) do { \
    e = hipFuncSetAttribute((const void *)nb_ring_kernel<N4_, M4_>, hipFuncAttributeMaxDynamicSharedMemorySize, cv.total); \
    if (e == hipSuccess) { \
      hipLaunchKernelGGL((nb_ring_kernel<N4_, M4_>), dim3(grid), dim3(NBR_THREADS), cv.total, stream, num, cat, rows, L, D, cv, ring, \
                         partials, skip, mask, kept); \
      e = hipGetLastError(); } } while (0)
#define ROW(N4_) switch ((L.m + 3) / 4) { case 1: GO(N4_, 4); break; case 2: GO(N4_, 8); break; case 3: GO(N4_, 12); break; \
                                           case 4: GO(N4_, 16); break; case 5: GO(N4_, 20); break; default: break; }
  switch ((L.n + 3) / 4) {
    case 0: ROW(0) break; case 1: ROW(4) break; case 2: ROW(8) break; case 3: ROW(12) break; case 4: ROW(16) break;
    case 5: ROW(20) break; default: break;
  }
#undef ROW
#undef GO
  if (e != hipSuccess) return e;
  if (ev1 && (e = hipEventRecord(ev1, stream)) != hipSuccess) return e;
  return launch_gram_fold(partials, grid, acc, stream);
}

}  // namespace cofactor
