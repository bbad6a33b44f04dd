// multiply_triple over two vectors of triples, in one pass per output row.
//
// Replaces Triple::multiply (duckdb_extension/src/triple/mul.cpp:97-107 lin, 262-289 quad, 185-217 /
// 377-446 / 542-598 the three list families) and its NB form (mul_nb.cpp:246-262): output row i =
// a[a_sel[i]] x b[b_sel[i]].
//
// Every output sub-list is one source sub-list scaled, or the outer product of two (A's keys of column
// c1 with B's keys of column c2, mul.cpp:564-580).  So a row's payload sizes follow from the TOTALS of
// its two sources' families:
//     lin_cat       L_A + L_B
//     quad_num_cat  Q_A + n_A L_B + n_B L_A + Q_B
//     quad_cat      C_A + C_B + L_A L_B
// (L, Q, C = entries of the triple's lin_cat / quad_num_cat / quad_cat lists).
//   mul_pair_len_kernel  one thread per row: the three totals (a scan per family over ROWS — not over
//                        rows x sub-lists, 30 times as many at 2_2 x 2_2 — gives the rows' places)
//   mul_fill_kernel      one wave per row: dense part by lanes; the row's sub-lists (all three families
//                        side by side, 64 at a time) get their places from a wave scan, their
//                        descriptors go to LDS, and the ENTRIES are then dealt to the lanes one each —
//                        consecutive lanes write consecutive entries of the output arrays.
#include "device.hpp"
#include "ring.hpp"

#include <cstdlib>

namespace cofactor {

namespace {

__host__ __device__ inline int mf_tri(int k) { return k * (k + 1) / 2; }
__host__ __device__ inline int mf_pair_q(int c1, int c2, int m) { return c1 * m - c1 * (c1 - 1) / 2 + (c2 - c1); }
__device__ inline void mf_pair_decode(int q, int m, int &c1, int &c2) {
  c1 = 0;
  while (q >= m - c1) { q -= m - c1; c1++; }
  c2 = c1 + q;
}

__device__ inline uint64_t mf_total(const uint64_t *outer, const uint64_t *sub, uint64_t i, int cnt) {
  if (cnt == 0) return 0;
  const uint64_t first = outer[2 * i];
  uint64_t t = 0;
  for (int s = 0; s < cnt; s++) t += sub[2 * (first + s) + 1];
  return t;
}

__global__ __launch_bounds__(256) void mul_pair_len_kernel(cofactor_tvec a, const uint32_t *__restrict__ asel, cofactor_tvec b,
                                                           const uint32_t *__restrict__ bsel, uint64_t rows,
                                                           uint64_t *__restrict__ t0, uint64_t *__restrict__ t1,
                                                           uint64_t *__restrict__ t2) {
  const int nA = a.n, mA = a.m, nB = b.n, mB = b.m, kind = a.kind;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t ia = asel ? asel[i] : i, ib = bsel ? bsel[i] : i;
    const uint64_t LA = mf_total(a.lc_outer, a.lc_sub, ia, mA), LB = mf_total(b.lc_outer, b.lc_sub, ib, mB);
    t0[i] = LA + LB;
    if (!kind) {
      const uint64_t QA = mf_total(a.nc_outer, a.nc_sub, ia, nA * mA), QB = mf_total(b.nc_outer, b.nc_sub, ib, nB * mB);
      const uint64_t CA = mf_total(a.cc_outer, a.cc_sub, ia, mf_tri(mA)), CB = mf_total(b.cc_outer, b.cc_sub, ib, mf_tri(mB));
      t1[i] = QA + (uint64_t)nA * LB + (uint64_t)nB * LA + QB;
      t2[i] = CA + CB + LA * LB;
    }
  }
}

// source of a sub-list
enum : int { SRC_A_LC = 0, SRC_B_LC, SRC_A_NC, SRC_B_NC, SRC_A_CC, SRC_B_CC, SRC_OUTER };

struct MulWaveLds {                 // one wave's sub-list descriptors (64 at a time): resolved pointers, so that
  const int32_t *sk1[64];           // an entry costs its loads, one multiply and its stores
  const int32_t *sk2[64];           // second key: the cc source's key2, or B's keys of an outer product
  const float *sv1[64];
  const float *sv2[64];             // B's values of an outer product, else null
  int32_t *dk1[64];                 // the sub-list's place in the output arrays
  int32_t *dk2[64];                 // (quad_cat only, else null)
  float *dv[64];
  unsigned rel[64];                 // entries of this block before the sub-list
  unsigned l2[64];                  // length of the second source (outer products), else 0
  unsigned magic[64];               // ceil(2^32 / l2): x / l2 = umulhi(x, magic) while x < 2^16
  float scale[64];
};

// exclusive scan inside groups of GL lanes (gl = lane within the group)
template <int GL>
__device__ __forceinline__ unsigned mf_group_excl_scan(unsigned v, int gl, unsigned &total) {
  unsigned incl = v;
#pragma unroll
  for (int d = 1; d < GL; d <<= 1) {
    const unsigned t = __shfl_up(incl, d, GL);
    if (gl >= d) incl += t;
  }
  total = __shfl(incl, GL - 1, GL);
  return incl - v;
}

// GL lanes per output row: 64, or 16 for small shapes — four rows per wave then, four times the loads
// in flight (with a whole wave per row of 30 sub-lists and 192 entries the kernel waited on its chain of
// dependent loads: selection -> outer entry -> sub-list entry -> payload, 2.7e8 rows/s).
template <int GL>
__global__ __launch_bounds__(256) void mul_fill_kernel(cofactor_tvec a, const uint32_t *__restrict__ asel, cofactor_tvec b,
                                                       const uint32_t *__restrict__ bsel, uint64_t rows,
                                                       const uint64_t *__restrict__ base0, const uint64_t *__restrict__ base1,
                                                       const uint64_t *__restrict__ base2, cofactor_tvec o) {
  __shared__ MulWaveLds l_all[4];
  constexpr int NG = 64 / GL;                          // rows a wave works on side by side
  const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63, grp = wl / GL, lane = wl - grp * GL, g0 = grp * GL;
  MulWaveLds &W = l_all[wave];
  const int nA = a.n, mA = a.m, nB = b.n, mB = b.m, nR = nA + nB, mR = mA + mB, kind = a.kind;
  const int TR = kind ? nR : mf_tri(nR), D = 1 + nR + TR;
  const int P0 = mR, P1 = kind ? 0 : nR * mR, P2 = kind ? 0 : mf_tri(mR), PT = P0 + P1 + P2;
  for (uint64_t ib0 = ((uint64_t)blockIdx.x * 4 + wave) * NG; ib0 < rows; ib0 += (uint64_t)gridDim.x * 4 * NG) {
    const bool live = ib0 + grp < rows;                // (rows past the end: the lanes only keep the wave's barriers company)
    const uint64_t i = live ? ib0 + grp : rows - 1;
    const uint64_t ia = asel ? asel[i] : i, ib = bsel ? bsel[i] : i;
    const float Na = (float)a.N[ia], Nb = (float)b.N[ib];
    const float *la = a.lin + a.lin_e[2 * ia], *lb = b.lin + b.lin_e[2 * ib];
    // ---- N, lin, quad and their list entries ------------------------------------------------------------
    {
      const float *qa = a.quad + a.quad_e[2 * ia], *qb = b.quad + b.quad_e[2 * ib];
      for (int e = lane; e < (live ? D : 0); e += GL) {
        if (e == 0) {
          o.N[i] = a.N[ia] * b.N[ib];                  // int32 product (mul.cpp:46-49)
          o.lin_e[2 * i] = i * nR; o.lin_e[2 * i + 1] = nR;
          o.quad_e[2 * i] = i * TR; o.quad_e[2 * i + 1] = TR;
          // a list family without sub-lists (no key columns / no numeric columns): empty outer lists
          if (mR == 0 && o.lc_outer) { o.lc_outer[2 * i] = 0; o.lc_outer[2 * i + 1] = 0; }
          if (!kind && nR * mR == 0 && o.nc_outer) { o.nc_outer[2 * i] = 0; o.nc_outer[2 * i + 1] = 0; }
          if (!kind && mR == 0 && o.cc_outer) { o.cc_outer[2 * i] = 0; o.cc_outer[2 * i + 1] = 0; }
        } else if (e <= nR) {                          // lin = [N_B lin_A | N_A lin_B] (mul.cpp:97-107)
          const int k = e - 1;
          o.lin[i * nR + k] = k < nA ? la[k] * Nb : lb[k - nA] * Na;
        } else {
          const int q = e - 1 - nR;
          float val;
          if (kind) val = q < nA ? qa[q] * Nb : qb[q - nA] * Na;        // mul_nb.cpp:246-262
          else {                                       // upper triangle of [[N_B Q_A, lin_A (x) lin_B], [., N_A Q_B]]
            int j = 0, r = q;
            while (r >= nR - j) { r -= nR - j; j++; }
            const int k = j + r;
            if (k < nA) val = qa[mf_pair_q(j, k, nA)] * Nb;
            else if (j < nA) val = la[j] * lb[k - nA];
            else val = qb[mf_pair_q(j - nA, k - nA, nB)] * Na;
          }
          o.quad[i * TR + q] = val;
        }
      }
    }
    if (PT == 0) continue;                               // (uniform)
    // ---- the list families ------------------------------------------------------------------------------
    uint64_t run0 = base0[i], run1 = kind ? 0 : base1[i], run2 = kind ? 0 : base2[i];
    const uint64_t fa_lc = mA ? a.lc_outer[2 * ia] : 0, fb_lc = mB ? b.lc_outer[2 * ib] : 0;
    for (int c0 = 0; c0 < PT; c0 += GL) {
      const int t = c0 + lane;
      const bool in = live && t < PT;
      const int f = t < P0 ? 0 : (t < P0 + P1 ? 1 : 2);
      const int s = t - (f == 0 ? 0 : (f == 1 ? P0 : P0 + P1));
      const int32_t *sk1 = nullptr, *sk2 = nullptr;
      const float *sv1 = nullptr, *sv2 = nullptr;
      unsigned l1 = 0, l2 = 0;
      float scale = 1.f;
      if (in) {
        const uint64_t *e1 = nullptr, *e2 = nullptr;
        int src = SRC_A_LC;
        if (f == 0) {                                  // lin_cat = [N_B lcat_A | N_A lcat_B]
          if (s < mA) { e1 = a.lc_sub + 2 * (fa_lc + s); scale = Nb; }
          else { e1 = b.lc_sub + 2 * (fb_lc + (s - mA)); src = SRC_B_LC; scale = Na; }
        } else if (f == 1) {                           // quad_num_cat, numeric-major over (A|B), key-minor over (A|B)
          const int j = s / mR, c = s - j * mR;
          if (j < nA && c < mA) { e1 = a.nc_sub + 2 * (a.nc_outer[2 * ia] + j * mA + c); src = SRC_A_NC; scale = Nb; }
          else if (j < nA) { e1 = b.lc_sub + 2 * (fb_lc + (c - mA)); src = SRC_B_LC; scale = la[j]; }
          else if (c < mA) { e1 = a.lc_sub + 2 * (fa_lc + c); scale = lb[j - nA]; }
          else { e1 = b.nc_sub + 2 * (b.nc_outer[2 * ib] + (j - nA) * mB + (c - mA)); src = SRC_B_NC; scale = Na; }
        } else {                                       // quad_cat over the joined key columns
          int c1, c2;
          mf_pair_decode(s, mR, c1, c2);
          if (c2 < mA) { e1 = a.cc_sub + 2 * (a.cc_outer[2 * ia] + mf_pair_q(c1, c2, mA)); src = SRC_A_CC; scale = Nb; }
          else if (c1 >= mA) { e1 = b.cc_sub + 2 * (b.cc_outer[2 * ib] + mf_pair_q(c1 - mA, c2 - mA, mB)); src = SRC_B_CC; scale = Na; }
          else {                                       // A x B: key-set outer product, count_A * count_B (mul.cpp:564-580)
            e1 = a.lc_sub + 2 * (fa_lc + c1);
            e2 = b.lc_sub + 2 * (fb_lc + (c2 - mA));
            src = SRC_OUTER;
          }
        }
        const uint64_t s1 = e1[0];
        l1 = (unsigned)e1[1];
        switch (src) {
          case SRC_A_LC: case SRC_OUTER: sk1 = a.lc_key + s1; sv1 = a.lc_val + s1; break;
          case SRC_B_LC: sk1 = b.lc_key + s1; sv1 = b.lc_val + s1; break;
          case SRC_A_NC: sk1 = a.nc_key + s1; sv1 = a.nc_val + s1; break;
          case SRC_B_NC: sk1 = b.nc_key + s1; sv1 = b.nc_val + s1; break;
          case SRC_A_CC: sk1 = a.cc_key1 + s1; sk2 = a.cc_key2 + s1; sv1 = a.cc_val + s1; break;
          default: sk1 = b.cc_key1 + s1; sk2 = b.cc_key2 + s1; sv1 = b.cc_val + s1; break;
        }
        if (e2) { const uint64_t s2 = e2[0]; l2 = (unsigned)e2[1]; sk2 = b.lc_key + s2; sv2 = b.lc_val + s2; }
      }
      const unsigned len = in ? (sv2 ? l1 * l2 : l1) : 0;
      unsigned total;
      const unsigned excl = mf_group_excl_scan<GL>(len, lane, total);
      // entries of this block that belong to the families before mine
      const int f1 = min(max(P0 - c0, 0), GL), f2 = min(max(P0 + P1 - c0, 0), GL);      // first lanes of families 1, 2
      const unsigned y1 = __shfl(excl, f1 < GL ? f1 : 0, GL), y2 = __shfl(excl, f2 < GL ? f2 : 0, GL);
      const unsigned x1 = f1 < GL ? y1 : total, x2 = f2 < GL ? y2 : total;
      const uint64_t place = f == 0 ? run0 + excl : (f == 1 ? run1 + (excl - x1) : run2 + (excl - x2));
      run0 += x1; run1 += x2 - x1; run2 += total - x2;
      if (in) {
        uint64_t *sub = f == 0 ? o.lc_sub : (f == 1 ? o.nc_sub : o.cc_sub);
        uint64_t *outer = f == 0 ? o.lc_outer : (f == 1 ? o.nc_outer : o.cc_outer);
        const uint64_t per = f == 0 ? P0 : (f == 1 ? P1 : P2), w = i * per + s;
        sub[2 * w] = place; sub[2 * w + 1] = len;
        if (s == 0) { outer[2 * i] = w; outer[2 * i + 1] = per; }
      }
      W.sk1[wl] = sk1; W.sk2[wl] = sk2; W.sv1[wl] = sv1; W.sv2[wl] = sv2;
      W.dk1[wl] = (f == 0 ? o.lc_key : (f == 1 ? o.nc_key : o.cc_key1)) + place;
      W.dk2[wl] = f == 2 ? o.cc_key2 + place : nullptr;
      W.dv[wl] = (f == 0 ? o.lc_val : (f == 1 ? o.nc_val : o.cc_val)) + place;
      W.l2[wl] = l2; W.scale[wl] = scale;
      W.magic[wl] = l2 > 1 ? 0xFFFFFFFFu / l2 + 1u : 0u;   // = ceil(2^32 / l2)
      W.rel[wl] = in ? excl : total;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // four entries per lane at a time: all their loads are on the way before the first store (one
      // entry per lane and round left the rounds waiting on each other's loads: 82 % of the wave cycles)
      constexpr int EU = 4;
      for (unsigned e0 = lane; e0 < total; e0 += EU * GL) {
        int32_t k1[EU], k2[EU];
        float val[EU];
        int32_t *d1[EU], *d2[EU];
        float *dv[EU];
#pragma unroll
        for (int u = 0; u < EU; u++) {
          const unsigned e = e0 + u * GL;
          d1[u] = nullptr; d2[u] = nullptr; dv[u] = nullptr; k1[u] = 0; k2[u] = 0; val[u] = 0.f;
          if (e < total) {
            int lo = g0;                               // last sub-list whose first entry is <= e (empty ones share their successor's)
#pragma unroll
            for (int step = GL / 2; step > 0; step >>= 1)
              if (W.rel[lo + step] <= e) lo += step;
            const unsigned x = e - W.rel[lo];
            const float *pv2 = W.sv2[lo];
            d1[u] = W.dk1[lo] + x; d2[u] = W.dk2[lo]; dv[u] = W.dv[lo] + x;
            if (d2[u]) d2[u] += x;
            if (pv2) {
              const unsigned n2 = W.l2[lo];
              const unsigned xa = n2 > 1 ? (x < 65536u ? __umulhi(x, W.magic[lo]) : x / n2) : x;
              const unsigned xb = x - xa * n2;
              k1[u] = W.sk1[lo][xa]; k2[u] = W.sk2[lo][xb]; val[u] = W.sv1[lo][xa] * pv2[xb];
            } else {
              k1[u] = W.sk1[lo][x];
              if (d2[u]) k2[u] = W.sk2[lo][x];
              val[u] = W.sv1[lo][x] * W.scale[lo];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < EU; u++) {
          if (d1[u]) {
            *d1[u] = k1[u];
            if (d2[u]) *d2[u] = k2[u];
            *dv[u] = val[u];
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
}

}  // namespace

hipError_t launch_mul_pair_lens(const cofactor_tvec &a, const uint32_t *asel, const cofactor_tvec &b, const uint32_t *bsel,
                                uint64_t rows, uint64_t *t0, uint64_t *t1, uint64_t *t2, hipStream_t stream) {
  if (rows == 0) return hipSuccess;
  const unsigned grid = (unsigned)std::min<uint64_t>((rows + 255) / 256, 8192);
  hipLaunchKernelGGL(mul_pair_len_kernel, dim3(grid), dim3(256), 0, stream, a, asel, b, bsel, rows, t0, t1, t2);
  return hipGetLastError();
}

hipError_t launch_mul_fill(const cofactor_tvec &a, const uint32_t *asel, const cofactor_tvec &b, const uint32_t *bsel,
                           uint64_t rows, const uint64_t *base0, const uint64_t *base1, const uint64_t *base2,
                           const cofactor_tvec &out, uint64_t entries, int cus, hipStream_t stream) {
  if (rows == 0) return hipSuccess;
  const int mR = a.m + b.m, nR = a.n + b.n;
  const int PT = mR + (a.kind ? 0 : nR * mR + mR * (mR + 1) / 2);
  static const int dev_gl = [] { const char *v = getenv("COFACTOR_MUL_GL"); return v ? atoi(v) : 0; }();
  // lanes per output row: 2_2 x 2_2 (30 sub-lists), 2e6 rows, 192 entries per row: 64 -> 7.3 ms, 32 -> 5.2,
  // 16 -> 4.3, 8 -> 3.8 (then four entries per lane in flight: 8 -> 3.4); 1 685 entries per row (16 keys per
  // column): 8 -> 22.7 ms, 16 -> 20.6, 32 -> 20.1, 64 -> 22.5
  const uint64_t per_row = entries / std::max<uint64_t>(rows, 1);
  int gl = PT <= 40 ? 8 : (PT <= 128 ? 16 : (PT <= 320 ? 32 : 64));
  if (per_row > 256) gl = std::max(gl, 16);
  if (per_row > 1024) gl = std::max(gl, 32);
  if (dev_gl) gl = dev_gl;
  if (gl == 4) {
    const unsigned grid = (unsigned)std::min<uint64_t>((rows + 63) / 64, (uint64_t)cus * 8);
    hipLaunchKernelGGL(mul_fill_kernel<4>, dim3(grid), dim3(256), 0, stream, a, asel, b, bsel, rows, base0, base1, base2, out);
  } else if (gl == 8) {
    const unsigned grid = (unsigned)std::min<uint64_t>((rows + 31) / 32, (uint64_t)cus * 8);
    hipLaunchKernelGGL(mul_fill_kernel<8>, dim3(grid), dim3(256), 0, stream, a, asel, b, bsel, rows, base0, base1, base2, out);
  } else if (gl == 32) {
    const unsigned grid = (unsigned)std::min<uint64_t>((rows + 7) / 8, (uint64_t)cus * 8);
    hipLaunchKernelGGL(mul_fill_kernel<32>, dim3(grid), dim3(256), 0, stream, a, asel, b, bsel, rows, base0, base1, base2, out);
  } else if (gl == 16) {
    const unsigned grid = (unsigned)std::min<uint64_t>((rows + 15) / 16, (uint64_t)cus * 8);
    hipLaunchKernelGGL(mul_fill_kernel<16>, dim3(grid), dim3(256), 0, stream, a, asel, b, bsel, rows, base0, base1, base2, out);
  } else {
    const unsigned grid = (unsigned)std::min<uint64_t>((rows + 3) / 4, (uint64_t)cus * 8);
    hipLaunchKernelGGL(mul_fill_kernel<64>, dim3(grid), dim3(256), 0, stream, a, asel, b, bsel, rows, base0, base1, base2, out);
  }
  return hipGetLastError();
}

}  // namespace cofactor
