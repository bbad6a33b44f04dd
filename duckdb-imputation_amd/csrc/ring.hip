// Batched ring operations on vectors of triples and the GROUP BY state pool:
//   lift_kernel            to_cofactor / to_nb_agg       (triple/lift.cpp:15-243, lift_to_nb_agg.cpp:13-136)
//   tvec_dense_kernel      sum_triple, dense children    (triple/sum/sum.cpp:86-149)
//   tvec_keys_kernel       sum_triple, key lists          (triple/sum/sum.cpp:197-260)
//   mul_*_kernel           multiply_triple / _nb_agg      (triple/mul.cpp:19-611, mul_nb.cpp:20-268)
//   groups_*_kernel        sum_to_triple ... GROUP BY g   (per-row state pointers, sum_no_lift.cpp:84-214)
// A vector of triples is held exactly as DuckDB holds the reference's result STRUCT (cofactor_tvec in
// include/cofactor_hip.h): one array per leaf, (offset, length) pairs per list level.
#include "ring.hpp"

#include <hipcub/hipcub.hpp>

namespace cofactor {

namespace {

__host__ __device__ inline int tri_i(int k) { return k * (k + 1) / 2; }
__host__ __device__ inline int pair_q(int c1, int c2, int m) { return c1 * m - c1 * (c1 - 1) / 2 + (c2 - c1); }
// q -> (c1, c2), c1 <= c2, of an m-column upper triangle
__device__ inline void pair_decode(int q, int m, int &c1, int &c2) {
  c1 = 0;
  while (q >= m - c1) { q -= m - c1; c1++; }
  c2 = c1 + q;
}

// ---- to_cofactor ---------------------------------------------------------------------------------
// A workgroup lifts a tile of 256 rows: the tile's inputs go to LDS once (every x_j is used
// 1 + n + m times), then each output array's segment of the tile — contiguous in memory — is
// written front to back by all threads, element e of the segment by thread e mod 256 (coalesced,
// 16-byte stores for the list entries).  All index arithmetic is 32-bit inside the tile (row = e /
// W through a multiply-high with W's reciprocal); the first version, one thread per element of
// the whole output with 64-bit divisions, was ALU-bound at 2.3-2.7 TB/s written.
constexpr int LIFT_ROWS = 256;
struct LiftDiv { unsigned mul; };                   // e / w == __umulhi(e, mul) for e * w < 2^32
__host__ __device__ inline LiftDiv lift_div(unsigned w) { return LiftDiv{w <= 1 ? 0u : (unsigned)(0x100000000ull / w) + 1u}; }
__device__ __forceinline__ unsigned lift_quot(unsigned e, unsigned w, LiftDiv d) { return w <= 1 ? e : __umulhi(e, d.mul); }

__global__ __launch_bounds__(256) void lift_kernel(NumCols num, CatCols cat, int n, int m, int kind, uint64_t rows,
                                                   cofactor_tvec o) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lift_lds[];
  const int T = kind ? n : tri_i(n), Tm = kind ? 0 : tri_i(m), nm = kind ? 0 : n * m;
  float *xs = reinterpret_cast<float *>(lift_lds);                       // [n][LIFT_ROWS]
  int32_t *ks = reinterpret_cast<int32_t *>(xs + n * LIFT_ROWS);         // [m][LIFT_ROWS]
  unsigned char *qa = reinterpret_cast<unsigned char *>(ks + m * LIFT_ROWS);   // quad entry q -> (j, k)
  unsigned char *qb = qa + 256;
  unsigned char *pa = qb + 256;                                          // quad_cat sub-list s -> (c1, c2)
  unsigned char *pb = pa + 256;
  const int tid = threadIdx.x;
  for (int q = tid; q < T; q += 256) {
    int j = 0, r = q, k;
    if (kind) { j = k = q; }
    else { while (r >= n - j) { r -= n - j; j++; } k = j + r; }
    qa[q] = (unsigned char)j; qb[q] = (unsigned char)k;
  }
  for (int s = tid; s < Tm; s += 256) {
    int c1, c2;
    pair_decode(s, m, c1, c2);
    pa[s] = (unsigned char)c1; pb[s] = (unsigned char)c2;
  }
  const LiftDiv dn = lift_div(n), dT = lift_div(T), dm = lift_div(m), dnm = lift_div(nm), dTm = lift_div(Tm);
  const uint64_t ntiles = (rows + LIFT_ROWS - 1) / LIFT_ROWS;
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint64_t row0 = tile * LIFT_ROWS;
    const unsigned R = (unsigned)min<uint64_t>(LIFT_ROWS, rows - row0);
    __syncthreads();                                 // (previous tile's readers done; tables written)
    for (int k = 0; k < n; k++)
      if ((unsigned)tid < R) xs[k * LIFT_ROWS + tid] = num.p[k][row0 + tid];
    for (int c = 0; c < m; c++)
      if ((unsigned)tid < R) ks[c * LIFT_ROWS + tid] = cat.p[c][row0 + tid];
    __syncthreads();
    if ((unsigned)tid < R) {                         // per-row items: N and the five outer list entries
      const uint64_t i = row0 + tid;
      o.N[i] = 1;
      reinterpret_cast<u64x2 *>(o.lin_e)[i] = u64x2{i * n, (unsigned long long)n};
      reinterpret_cast<u64x2 *>(o.quad_e)[i] = u64x2{i * T, (unsigned long long)T};
      if (o.lc_outer) reinterpret_cast<u64x2 *>(o.lc_outer)[i] = u64x2{i * m, (unsigned long long)m};
      if (!kind && o.nc_outer) reinterpret_cast<u64x2 *>(o.nc_outer)[i] = u64x2{i * nm, (unsigned long long)nm};
      if (!kind && o.cc_outer) reinterpret_cast<u64x2 *>(o.cc_outer)[i] = u64x2{i * Tm, (unsigned long long)Tm};
    }
    {                                                // lin: [R][n]
      float *dst = o.lin + row0 * n;
      for (unsigned e = tid; e < R * n; e += 256) {
        const unsigned r = lift_quot(e, n, dn), k = e - r * n;
        dst[e] = xs[k * LIFT_ROWS + r];
      }
    }
    {                                                // quad: [R][T], float products as the reference stores them (lift.cpp:119-136)
      float *dst = o.quad + row0 * T;
      for (unsigned e = tid; e < R * T; e += 256) {
        const unsigned r = lift_quot(e, T, dT), q = e - r * T;
        dst[e] = xs[qa[q] * LIFT_ROWS + r] * xs[qb[q] * LIFT_ROWS + r];
      }
    }
    if (m > 0) {                                     // lin_cat: [{key, 1}] per key column (lift.cpp:94-105)
      const uint64_t u0 = row0 * m;
      for (unsigned e = tid; e < R * m; e += 256) {
        const unsigned r = lift_quot(e, m, dm), c = e - r * m;
        reinterpret_cast<u64x2 *>(o.lc_sub)[u0 + e] = u64x2{u0 + e, 1ull};
        o.lc_key[u0 + e] = ks[c * LIFT_ROWS + r];
        o.lc_val[u0 + e] = 1.f;
      }
    }
    if (!kind && nm > 0) {                           // quad_num_cat[(j m + c)] = [{key_c, x_j}] (lift.cpp:156-176)
      const uint64_t u0 = row0 * nm;
      for (unsigned e = tid; e < R * nm; e += 256) {
        const unsigned r = lift_quot(e, nm, dnm), s = e - r * nm;
        const unsigned j = lift_quot(s, m, dm), c = s - j * m;
        reinterpret_cast<u64x2 *>(o.nc_sub)[u0 + e] = u64x2{u0 + e, 1ull};
        o.nc_key[u0 + e] = ks[c * LIFT_ROWS + r];
        o.nc_val[u0 + e] = xs[j * LIFT_ROWS + r];
      }
    }
    if (!kind && Tm > 0) {                           // quad_cat[(c1, c2 >= c1)] = [{k1, k2, 1}] (lift.cpp:199-219)
      const uint64_t u0 = row0 * Tm;
      for (unsigned e = tid; e < R * Tm; e += 256) {
        const unsigned r = lift_quot(e, Tm, dTm), s = e - r * Tm;
        reinterpret_cast<u64x2 *>(o.cc_sub)[u0 + e] = u64x2{u0 + e, 1ull};
        o.cc_key1[u0 + e] = ks[pa[s] * LIFT_ROWS + r];
        o.cc_key2[u0 + e] = ks[pb[s] * LIFT_ROWS + r];
        o.cc_val[u0 + e] = 1.f;
      }
    }
  }
}

// ---- sum_triple: dense children -------------------------------------------------------------------
// acc[0] += sum N, acc[1 + k] += sum lin[.][k], acc[1 + n + q] += sum quad[.][q].
// One child array per launch (width W values per row, row i at child[entries[2 i] ..]): thread t of
// a block owns column t % W of the rows t / W, t / W + RPB, .. of the block's chunk, so that a
// block iteration reads RPB = 256 / W whole rows = one contiguous stretch (coalesced), four
// iterations in flight; the per-thread sums meet in LDS.
__global__ __launch_bounds__(256) void tvec_dense_kernel(const float *__restrict__ child, const uint64_t *__restrict__ entries,
                                                         const int32_t *__restrict__ Ncol, uint64_t count, int W,
                                                         double *__restrict__ acc) {
  __shared__ double red[256];
  const int tid = threadIdx.x;
  const int RPB = 256 / W;                            // rows per block iteration (W <= 256)
  const int k = tid % W, r = tid / W;
  const uint64_t per = (count + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = (uint64_t)blockIdx.x * per, hi = min(count, lo + per);
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  if (r < RPB) {
    uint64_t i = lo + r;
    if (Ncol) {                                       // the INTEGER column N (W = 1)
      for (; i + 3 * (uint64_t)RPB < hi; i += 4 * (uint64_t)RPB) {
        s0 += (double)Ncol[i]; s1 += (double)Ncol[i + RPB]; s2 += (double)Ncol[i + 2 * RPB]; s3 += (double)Ncol[i + 3 * RPB];
      }
      for (; i < hi; i += RPB) s0 += (double)Ncol[i];
    } else {
      for (; i + 3 * (uint64_t)RPB < hi; i += 4 * (uint64_t)RPB) {
        const float a = child[entries[2 * i] + k], b = child[entries[2 * (i + RPB)] + k];
        const float c = child[entries[2 * (i + 2 * RPB)] + k], d = child[entries[2 * (i + 3 * RPB)] + k];
        s0 += (double)a; s1 += (double)b; s2 += (double)c; s3 += (double)d;
      }
      for (; i < hi; i += RPB) s0 += (double)child[entries[2 * i] + k];
    }
  }
  red[tid] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (tid < W) {
    double v = 0;
    for (int rr = 0; rr < RPB; rr++) v += red[rr * W + tid];
    if (lo < hi) unsafeAtomicAdd(&acc[tid], v);
  }
}

// acc image of the aggregate (gram.hip layout) += the reduced dense vector; kept += N
__global__ void tvec_dense_apply_kernel(const double *__restrict__ red, int n, int kind, double *__restrict__ acc,
                                        unsigned long long *__restrict__ kept) {
  const int T = kind ? n : tri_i(n), i = threadIdx.x;
  if (i == 0) { *kept += (unsigned long long)(red[0] + 0.5); return; }
  if (i <= n) { acc[gram_lin_pos(i - 1, n)] += red[i]; return; }
  if (i < 1 + n + T) {
    int q = i - 1 - n, j = 0;
    if (kind) { acc[gram_quad_pos(q, q, n)] += red[i]; return; }
    while (q >= n - j) { q -= n - j; j++; }
    acc[gram_quad_pos(j, j + q, n)] += red[i];
  }
}

// ---- sum_triple: key lists --------------------------------------------------------------------------
// pass 0: every key of every lin_cat sub-list into its column's dictionary
// pass 1: lin_cat values -> cnt, quad_num_cat values -> s, quad_cat values -> p (codes via the dictionaries)
// One work item = one sub-list (m + n m + T(m) of them per triple; one entry each for lifted rows,
// up to #keys for grouped ones).
__global__ __launch_bounds__(256) void tvec_keys_kernel(cofactor_tvec v, CatLayout L, CatDevice D, int pass) {
  const int n = v.n, m = v.m;
  const int Tm = v.kind ? 0 : tri_i(m), nm = v.kind ? 0 : n * m;
  const uint64_t per_row = (uint64_t)m + (pass ? nm + Tm : 0);
  const uint64_t total = v.count * per_row;
  for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t i = w / per_row;
    int s = (int)(w % per_row);
    if (s < m) {
      const int c = s;
      const uint64_t sub = v.lc_outer[2 * i] + c, off = v.lc_sub[2 * sub], len = v.lc_sub[2 * sub + 1];
      for (uint64_t e = off; e < off + len; e++) {
        if (!pass) cat_dict_insert(D.ht_slot + L.ht_off[c], L.ht_cap[c], v.lc_key[e], D.flags);
        else {
          const int code = cat_lookup_code(D.ht_slot + L.ht_off[c], D.ht_code + L.ht_off[c], L.ht_cap[c], v.lc_key[e]);
          if (code < 0 || code >= L.kc[c]) { D.flags[1] = 1; continue; }
          atomicAdd(&D.cnt[L.cnt_off[c] + code], (unsigned long long)(v.lc_val[e] + 0.5f));
        }
      }
      continue;
    }
    s -= m;
    if (s < nm) {
      const int k = s / m, c = s % m;
      const uint64_t sub = v.nc_outer[2 * i] + s, off = v.nc_sub[2 * sub], len = v.nc_sub[2 * sub + 1];
      for (uint64_t e = off; e < off + len; e++) {
        const int code = cat_lookup_code(D.ht_slot + L.ht_off[c], D.ht_code + L.ht_off[c], L.ht_cap[c], v.nc_key[e]);
        if (code < 0 || code >= L.kc[c]) { D.flags[1] = 1; continue; }
        unsafeAtomicAdd(&D.s[L.s_off[c] + (long long)code * n + k], (double)v.nc_val[e]);
      }
      continue;
    }
    s -= nm;
    {
      int c1, c2;
      if (pair_is_sparse(L, s)) continue;             // (kept as a sorted list: tvec_sparse_fill_kernel)
      pair_decode(s, m, c1, c2);
      const uint64_t sub = v.cc_outer[2 * i] + s, off = v.cc_sub[2 * sub], len = v.cc_sub[2 * sub + 1];
      for (uint64_t e = off; e < off + len; e++) {
        const int k1 = cat_lookup_code(D.ht_slot + L.ht_off[c1], D.ht_code + L.ht_off[c1], L.ht_cap[c1], v.cc_key1[e]);
        const int k2 = cat_lookup_code(D.ht_slot + L.ht_off[c2], D.ht_code + L.ht_off[c2], L.ht_cap[c2], v.cc_key2[e]);
        if (k1 < 0 || k2 < 0 || k1 >= L.kc[c1] || k2 >= L.kc[c2]) { D.flags[1] = 1; continue; }
        atomicAdd(&D.p[L.p_off[s] + (long long)k1 * L.kc[c2] + k2], (unsigned long long)(v.cc_val[e] + 0.5f));
      }
    }
  }
}

// Pass 1 when the three tables (as doubles) and the dictionaries fit LDS: tables private to the
// workgroup (lifted rows hammer a handful of cells: global atomics on them serialise), added to
// the aggregate's at the end; dictionaries probed in LDS; FOUR sub-lists in flight per thread —
// the chain outer entry -> sub-list entry -> key / value -> probe is all dependent loads, and one
// chain per thread leaves the kernel waiting on memory latency (1.7e9 triples/s at 4_2 before).
// The three children go one after the other (KIND 0 lin_cat, 1 quad_num_cat, 2 quad_cat), so
// every loop is over sub-lists of one shape.
template <int KIND>
__device__ __forceinline__ bool tvec_child_to_lds(const cofactor_tvec &v, const CatLayout &L, double *l_t,
                                                  const unsigned long long *l_slot, const int32_t *l_code) {
  const int n = v.n, m = v.m;
  const uint64_t *outer = KIND == 0 ? v.lc_outer : (KIND == 1 ? v.nc_outer : v.cc_outer);
  const uint64_t *sube = KIND == 0 ? v.lc_sub : (KIND == 1 ? v.nc_sub : v.cc_sub);
  const int32_t *key1 = KIND == 0 ? v.lc_key : (KIND == 1 ? v.nc_key : v.cc_key1);
  const int32_t *key2 = v.cc_key2;
  const float *vals = KIND == 0 ? v.lc_val : (KIND == 1 ? v.nc_val : v.cc_val);
  const int per_row = KIND == 0 ? m : (KIND == 1 ? n * m : tri_i(m));
  const uint64_t total = v.count * (uint64_t)per_row;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  constexpr int U = 4;
  bool bad = false;
  for (uint64_t w0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w0 < total; w0 += U * stride) {
    int sidx[U];
    uint64_t off[U], len[U], sub[U], maxlen = 0;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint64_t w = w0 + u * stride;
      const bool in = w < total;
      const uint64_t i = in ? w / per_row : 0;
      sidx[u] = in ? (int)(w - i * per_row) : 0;
      sub[u] = in ? outer[2 * i] + sidx[u] : 0;
      len[u] = in ? 1 : 0;                        // (marks the slot; the real length follows)
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (KIND == 2 && len[u] && pair_is_sparse(L, sidx[u])) len[u] = 0;   // (kept as a sorted list)
      off[u] = len[u] ? sube[2 * sub[u]] : 0;
      len[u] = len[u] ? sube[2 * sub[u] + 1] : 0;
      maxlen = max(maxlen, len[u]);
    }
    for (uint64_t e = 0; e < maxlen; e++) {
      int32_t ka[U], kb[U];
      float val[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const bool live = e < len[u];
        const uint64_t at = off[u] + e;
        ka[u] = live ? key1[at] : 0;
        kb[u] = (KIND == 2 && live) ? key2[at] : 0;
        val[u] = live ? vals[at] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (e < len[u]) {
          int cell = -1;
          if (KIND == 0) {
            const int c = sidx[u];
            const int code = cat_lookup_code(l_slot + L.ht_off[c], l_code + L.ht_off[c], L.ht_cap[c], ka[u]);
            if (code >= 0 && code < L.kc[c]) cell = L.cnt_off[c] + code;
          } else if (KIND == 1) {
            const int k = sidx[u] / m, c = sidx[u] - k * m;
            const int code = cat_lookup_code(l_slot + L.ht_off[c], l_code + L.ht_off[c], L.ht_cap[c], ka[u]);
            if (code >= 0 && code < L.kc[c]) cell = L.n_cnt + L.s_off[c] + code * n + k;
          } else {
            int c1, c2;
            pair_decode(sidx[u], m, c1, c2);
            const int k1 = cat_lookup_code(l_slot + L.ht_off[c1], l_code + L.ht_off[c1], L.ht_cap[c1], ka[u]);
            const int k2 = cat_lookup_code(l_slot + L.ht_off[c2], l_code + L.ht_off[c2], L.ht_cap[c2], kb[u]);
            if (k1 >= 0 && k2 >= 0 && k1 < L.kc[c1] && k2 < L.kc[c2]) cell = L.n_cnt + L.n_s + L.p_off[sidx[u]] + k1 * L.kc[c2] + k2;
          }
          if (cell >= 0) unsafeAtomicAdd(&l_t[cell], (double)val[u]);
          else bad = true;
        }
      }
    }
  }
  return bad;
}

__global__ __launch_bounds__(256) void tvec_keys_lds_kernel(cofactor_tvec v, CatLayout L, CatDevice D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int cells = L.n_cnt + L.n_s + L.n_p;
  double *l_t = reinterpret_cast<double *>(lds_raw);                        // [cnt | s | p]
  unsigned long long *l_slot = reinterpret_cast<unsigned long long *>(l_t + cells);
  int32_t *l_code = reinterpret_cast<int32_t *>(l_slot + L.n_slots);
  for (int i = threadIdx.x; i < cells; i += 256) l_t[i] = 0.0;
  for (int i = threadIdx.x; i < L.n_slots; i += 256) { l_slot[i] = D.ht_slot[i]; l_code[i] = D.ht_code[i]; }
  __syncthreads();
  bool bad = tvec_child_to_lds<0>(v, L, l_t, l_slot, l_code);
  if (!v.kind) {
    bad = tvec_child_to_lds<1>(v, L, l_t, l_slot, l_code) || bad;
    bad = tvec_child_to_lds<2>(v, L, l_t, l_slot, l_code) || bad;
  }
  if (bad) D.flags[1] = 1;
  __syncthreads();
  for (int i = threadIdx.x; i < cells; i += 256) {
    const double t = l_t[i];
    if (t == 0.0) continue;
    if (i < L.n_cnt) atomicAdd(&D.cnt[i], (unsigned long long)(t + 0.5));
    else if (i < L.n_cnt + L.n_s) unsafeAtomicAdd(&D.s[i - L.n_cnt], t);
    else atomicAdd(&D.p[i - L.n_cnt - L.n_s], (unsigned long long)(t + 0.5));
  }
}

// ---- GROUP BY state pool ---------------------------------------------------------------------------
// Row g of tab (Dtot doubles): [N | lin(n) | quad(T) | cnt(n_cnt) | s(n_s) | p(n_p)], L's offsets.
// A wave takes 64 consecutive rows: lane r stages row r (group, values, key codes) in LDS, then the
// wave walks the rows and its lanes walk the row's cells, one double atomic each (a row's dense
// cells are contiguous in its group's table row).
constexpr int GR_THREADS = 256;

__global__ __launch_bounds__(GR_THREADS) void groups_accumulate_kernel(const int32_t *__restrict__ gid, NumCols num,
                                                                       CatCols cat, uint64_t rows, CatLayout L, CatDevice D,
                                                                       CatLayout Lg, CatDevice Dg, int is_key,
                                                                       double *__restrict__ tab, long long dtot) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int n = L.n, m = L.m, kind = L.kind;
  const int T = kind ? n : tri_i(n), Dd = 1 + n + T;
  const int ncat = m + (kind ? 0 : n * m + tri_i(m));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int stride = n + m + 1;                      // floats / ints per staged row
  float *xs = reinterpret_cast<float *>(lds_raw) + (size_t)wave * 64 * stride;
  int *is = reinterpret_cast<int *>(xs);
  const uint64_t nchunks = (rows + 63) / 64;
  for (uint64_t ch = (uint64_t)blockIdx.x * (GR_THREADS / 64) + wave; ch < nchunks; ch += (uint64_t)gridDim.x * (GR_THREADS / 64)) {
    const uint64_t r = ch * 64 + lane;
    const int valid = r < rows;
    if (valid) {
      int g = gid[r];
      if (is_key) g = cat_lookup_code(Dg.ht_slot, Dg.ht_code, Lg.ht_cap[0], g);
      is[lane * stride] = g;
      for (int k = 0; k < n; k++) xs[lane * stride + 1 + k] = num.p[k][r];
      for (int c = 0; c < m; c++)
        is[lane * stride + 1 + n + c] = cat_lookup_code(D.ht_slot + L.ht_off[c], D.ht_code + L.ht_off[c], L.ht_cap[c], cat.p[c][r]);
    }
    const int nvalid = (int)min<uint64_t>(64, rows - ch * 64);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // narrow triples: W = the cell count rounded up to a power of two lanes per row, 64 / W rows
    // side by side (15 cells at 2_2 would leave 49 of the 64 lanes idle)
    const int ncell = Dd + ncat;
    int W = 64;
    while (W / 2 >= ncell && W > 1) W >>= 1;
    const int RP = 64 / W, lrow = lane / W, lcell = lane - lrow * W;
    for (int rr = lrow; rr < nvalid; rr += RP) {
      const float *x = xs + rr * stride + 1;
      const int *cd = is + rr * stride + 1 + n;
      const int g = is[rr * stride];
      if (g < 0) { if (lcell == 0) D.flags[1] = 1; continue; }
      double *row = tab + (long long)g * dtot;
      for (int cell = lcell; cell < ncell; cell += W) {
        long long idx = cell;
        double val;
        if (cell == 0) val = 1.0;
        else if (cell <= n) val = (double)x[cell - 1];
        else if (cell < Dd) {
          int q = cell - 1 - n, j = 0, k;
          if (kind) j = k = q;
          else { while (q >= n - j) { q -= n - j; j++; } k = j + q; }
          val = (double)(x[j] * x[k]);               // float product (sum_no_lift.cpp:139)
        } else {
          int u = cell - Dd;
          if (u < m) {
            if (cd[u] < 0) { D.flags[1] = 1; continue; }
            idx = Dd + L.cnt_off[u] + cd[u]; val = 1.0;
          } else if (u < m + n * m) {
            u -= m;
            const int c = u / n, k = u % n;
            if (cd[c] < 0) continue;
            idx = (long long)Dd + L.n_cnt + L.s_off[c] + (long long)cd[c] * n + k; val = (double)x[k];
          } else {
            int c1, c2;
            pair_decode(u - m - n * m, m, c1, c2);
            if (cd[c1] < 0 || cd[c2] < 0) continue;
            idx = (long long)Dd + L.n_cnt + L.n_s + L.p_off[pair_q(c1, c2, m)] + (long long)cd[c1] * L.kc[c2] + cd[c2];
            val = 1.0;
          }
        }
        unsafeAtomicAdd(&row[idx], val);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// keys of a batch into the dictionaries: group keys (Lg / Dg, one column) and the key columns
__global__ __launch_bounds__(256) void groups_insert_kernel(const int32_t *__restrict__ gid, CatCols cat, uint64_t rows,
                                                            CatLayout L, CatDevice D, CatLayout Lg, CatDevice Dg, int is_key) {
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (uint64_t)gridDim.x * blockDim.x) {
    if (is_key) cat_dict_insert(Dg.ht_slot, Lg.ht_cap[0], gid[r], Dg.flags);
    for (int c = 0; c < L.m; c++) cat_dict_insert(D.ht_slot + L.ht_off[c], L.ht_cap[c], cat.p[c][r], D.flags);
  }
}

__global__ __launch_bounds__(256) void max_i32_kernel(const int32_t *__restrict__ v, uint64_t rows, int *__restrict__ out) {
  int mx = -1;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (uint64_t)gridDim.x * blockDim.x) mx = max(mx, v[r]);
  for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_down(mx, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, mx);
}

// table rows from one key-column layout to another (code capacities grew): cells keep their codes
__global__ __launch_bounds__(256) void groups_relayout_kernel(CatLayout Lo, CatLayout Ln, const double *__restrict__ to,
                                                              double *__restrict__ tn, long long dto, long long dtn, long long groups) {
  const int n = Lo.n, m = Lo.m, kind = Lo.kind;
  const int T = kind ? n : tri_i(n), Dd = 1 + n + T;
  for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < groups * dto; w += (long long)gridDim.x * blockDim.x) {
    const long long g = w / dto;
    const int cell = (int)(w % dto);
    long long idx;
    if (cell < Dd) idx = cell;
    else if (cell < Dd + Lo.n_cnt) {
      const int u = cell - Dd;
      int c = 0;
      while (c + 1 < m && u >= Lo.cnt_off[c + 1]) c++;
      idx = Dd + Ln.cnt_off[c] + (u - Lo.cnt_off[c]);
    } else if (cell < Dd + Lo.n_cnt + Lo.n_s) {
      const int u = cell - Dd - Lo.n_cnt;
      int c = 0;
      while (c + 1 < m && u >= Lo.s_off[c + 1]) c++;
      idx = (long long)Dd + Ln.n_cnt + Ln.s_off[c] + (u - Lo.s_off[c]);
    } else {
      const int u = cell - Dd - Lo.n_cnt - Lo.n_s;
      const int npairs = tri_i(m);
      int q = 0;
      while (q + 1 < npairs && u >= Lo.p_off[q + 1]) q++;
      int c1, c2;
      pair_decode(q, m, c1, c2);
      const int local = u - Lo.p_off[q];
      idx = (long long)Dd + Ln.n_cnt + Ln.n_s + Ln.p_off[q] + (long long)(local / Lo.kc[c2]) * Ln.kc[c2] + local % Lo.kc[c2];
    }
    tn[g * dtn + idx] = to[w];
  }
}

__global__ __launch_bounds__(256) void groups_combine_kernel(double *__restrict__ tab, long long dtot, long long dst, long long src) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < dtot; i += (long long)gridDim.x * blockDim.x)
    tab[dst * dtot + i] += tab[src * dtot + i];
}

// Groups -> a vector of triples (SumStateFinalize's layout, sum_state.cpp:116-464).  gorder[rank] = group
// table row; ord[cnt_off[c] + r] = the code holding column c's r-th smallest key (-1 past the
// last), keyof[cnt_off[c] + code] = its key.  mode 0: sub-list lengths; mode 1: fill.
__global__ __launch_bounds__(256) void groups_lists_kernel(const double *__restrict__ tab, long long dtot, CatLayout L,
                                                           const int32_t *__restrict__ gorder, long long groups,
                                                           const int32_t *__restrict__ ord, const int32_t *__restrict__ keyof,
                                                           int family, uint64_t *__restrict__ len,
                                                           const uint64_t *__restrict__ offs, cofactor_tvec o, int mode) {
  const int n = L.n, m = L.m, kind = L.kind;
  const int T = kind ? n : tri_i(n), Dd = 1 + n + T;
  const int per_row = family == 0 ? m : (family == 1 ? n * m : tri_i(m));
  for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < groups * per_row; w += (long long)gridDim.x * blockDim.x) {
    const long long i = w / per_row;
    const int s = (int)(w % per_row);
    const double *row = tab + (long long)gorder[i] * dtot;
    uint64_t cntv = 0;
    const uint64_t off = mode ? offs[w] : 0;
    if (family < 2) {
      const int c = family == 0 ? s : s % m, k = family == 0 ? 0 : s / m;
      const double *cnt = row + Dd + L.cnt_off[c];
      const double *sum = row + Dd + L.n_cnt + L.s_off[c];
      for (int r = 0; r < L.kc[c]; r++) {
        const int code = ord[L.cnt_off[c] + r];
        if (code < 0) break;
        if (cnt[code] == 0.0) continue;              // key known to the dictionary, not to this group
        if (mode) {
          if (family == 0) { o.lc_key[off + cntv] = keyof[L.cnt_off[c] + code]; o.lc_val[off + cntv] = (float)cnt[code]; }
          else { o.nc_key[off + cntv] = keyof[L.cnt_off[c] + code]; o.nc_val[off + cntv] = (float)sum[(long long)code * n + k]; }
        }
        cntv++;
      }
    } else {
      int c1, c2;
      pair_decode(s, m, c1, c2);
      const double *p = row + Dd + L.n_cnt + L.n_s + L.p_off[s];
      for (int r1 = 0; r1 < L.kc[c1]; r1++) {
        const int k1 = ord[L.cnt_off[c1] + r1];
        if (k1 < 0) break;
        for (int r2 = 0; r2 < L.kc[c2]; r2++) {
          const int k2 = ord[L.cnt_off[c2] + r2];
          if (k2 < 0) break;
          const double v = p[(long long)k1 * L.kc[c2] + k2];
          if (v == 0.0) continue;
          if (mode) {
            o.cc_key1[off + cntv] = keyof[L.cnt_off[c1] + k1]; o.cc_key2[off + cntv] = keyof[L.cnt_off[c2] + k2];
            o.cc_val[off + cntv] = (float)v;
          }
          cntv++;
        }
      }
    }
    if (!mode) { len[w] = cntv; continue; }
    uint64_t *sub = family == 0 ? o.lc_sub : (family == 1 ? o.nc_sub : o.cc_sub);
    uint64_t *outer = family == 0 ? o.lc_outer : (family == 1 ? o.nc_outer : o.cc_outer);
    sub[2 * w] = off; sub[2 * w + 1] = cntv;
    if (s == 0) { outer[2 * i] = (uint64_t)w; outer[2 * i + 1] = (uint64_t)per_row; }
  }
}

__global__ __launch_bounds__(256) void groups_dense_kernel(const double *__restrict__ tab, long long dtot, int n, int T,
                                                           const int32_t *__restrict__ gorder, long long groups, cofactor_tvec o) {
  const int m = o.m, kind = o.kind;
  const int D = 1 + n + T;
  for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < groups * D; w += (long long)gridDim.x * blockDim.x) {
    const long long i = w / D;
    const int e = (int)(w % D);
    const double v = tab[(long long)gorder[i] * dtot + e];
    if (e == 0) {
      o.N[i] = (int32_t)v;
      o.lin_e[2 * i] = (uint64_t)i * n; o.lin_e[2 * i + 1] = n;
      o.quad_e[2 * i] = (uint64_t)i * T; o.quad_e[2 * i + 1] = T;
      if (m == 0 && o.lc_outer) { o.lc_outer[2 * i] = 0; o.lc_outer[2 * i + 1] = 0; }
      if (!kind && n * m == 0 && o.nc_outer) { o.nc_outer[2 * i] = 0; o.nc_outer[2 * i + 1] = 0; }
      if (!kind && m == 0 && o.cc_outer) { o.cc_outer[2 * i] = 0; o.cc_outer[2 * i + 1] = 0; }
    } else if (e <= n) o.lin[i * n + (e - 1)] = (float)v;
    else o.quad[i * T + (e - 1 - n)] = (float)v;
  }
}

int grid_for(uint64_t items) {
  const uint64_t b = (items + 255) / 256;
  return (int)std::min<uint64_t>(std::max<uint64_t>(b, 1), 8192);
}

}  // namespace

hipError_t launch_lift(const NumCols &num, const CatCols &cat, int n, int m, int kind, uint64_t rows,
                       const cofactor_tvec &out, hipStream_t stream) {
  if (rows == 0) return hipSuccess;
  const size_t lds = (size_t)(n + m) * LIFT_ROWS * 4 + 1024;
  const uint64_t ntiles = (rows + LIFT_ROWS - 1) / LIFT_ROWS;
  const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, 8192);
  hipLaunchKernelGGL(lift_kernel, dim3(grid), dim3(256), lds, stream, num, cat, n, m, kind, rows, out);
  return hipGetLastError();
}

hipError_t launch_tvec_dense(const cofactor_tvec &v, double *red, double *acc, unsigned long long *kept, int grid,
                             hipStream_t stream) {
  if (v.count == 0) return hipSuccess;
  const int T = v.kind ? v.n : tri_i(v.n);
  hipError_t e = hipMemsetAsync(red, 0, sizeof(double) * 256, stream);
  if (e != hipSuccess) return e;
  const unsigned blocks = (unsigned)std::min<uint64_t>((v.count + 1023) / 1024, (uint64_t)grid);
  hipLaunchKernelGGL(tvec_dense_kernel, dim3(blocks), dim3(256), 0, stream, (const float *)nullptr, (const uint64_t *)nullptr,
                     v.N, v.count, 1, red);
  if (v.n > 0) {
    hipLaunchKernelGGL(tvec_dense_kernel, dim3(blocks), dim3(256), 0, stream, v.lin, v.lin_e, (const int32_t *)nullptr, v.count,
                       v.n, red + 1);
    hipLaunchKernelGGL(tvec_dense_kernel, dim3(blocks), dim3(256), 0, stream, v.quad, v.quad_e, (const int32_t *)nullptr, v.count,
                       T, red + 1 + v.n);
  }
  hipLaunchKernelGGL(tvec_dense_apply_kernel, dim3(1), dim3(256), 0, stream, red, v.n, v.kind, acc, kept);
  return hipGetLastError();
}

// quad_cat entries of the column pairs a state keeps as sorted lists: counted per pair, then written
// as (packed key pair, count) — in any order, sparse_merge_lists sorts and folds them
__global__ __launch_bounds__(256) void tvec_sparse_kernel(cofactor_tvec v, CatLayout L, unsigned long long *__restrict__ counts,
                                                          const unsigned long long *__restrict__ base,
                                                          unsigned long long *__restrict__ keys,
                                                          unsigned long long *__restrict__ cnt, int fill) {
  const int per = tri_i(v.m);
  const uint64_t total = v.count * (uint64_t)per;
  for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t i = w / per;
    const int q = (int)(w - i * per);
    if (!pair_is_sparse(L, q)) continue;
    const uint64_t sub = v.cc_outer[2 * i] + q, off = v.cc_sub[2 * sub], len = v.cc_sub[2 * sub + 1];
    if (len == 0) continue;
    const unsigned long long at = atomicAdd(&counts[q], (unsigned long long)len);
    if (!fill) continue;
    for (uint64_t e = 0; e < len; e++) {
      const unsigned long long k = ((unsigned long long)((uint32_t)v.cc_key1[off + e] ^ 0x80000000u) << 32) |
                                   (unsigned long long)((uint32_t)v.cc_key2[off + e] ^ 0x80000000u);   // = sparse_pack
      keys[base[q] + at + e] = k;
      cnt[base[q] + at + e] = (unsigned long long)(v.cc_val[off + e] + 0.5f);
    }
  }
}

hipError_t launch_tvec_sparse(const cofactor_tvec &v, const CatLayout &L, unsigned long long *counts,
                              const unsigned long long *base, unsigned long long *keys, unsigned long long *cnt, int fill,
                              hipStream_t stream) {
  const uint64_t total = v.count * (uint64_t)tri_i(v.m);
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(tvec_sparse_kernel, dim3(grid_for(total)), dim3(256), 0, stream, v, L, counts, base, keys, cnt, fill);
  return hipGetLastError();
}

hipError_t launch_tvec_keys(const cofactor_tvec &v, const CatLayout &L, const CatDevice &D, int pass, hipStream_t stream) {
  if (v.count == 0 || v.m == 0) return hipSuccess;
  const int Tm = v.kind ? 0 : tri_i(v.m), nm = v.kind ? 0 : v.n * v.m;
  const uint64_t total = v.count * ((uint64_t)v.m + (pass ? nm + Tm : 0));
  const size_t lds = (size_t)(L.n_cnt + L.n_s + L.n_p) * 8 + (size_t)L.n_slots * 12;
  if (pass == 1 && lds <= 40 * 1024) {
    const int grid = (int)std::min<uint64_t>((total + 1023) / 1024, 4096);
    hipLaunchKernelGGL(tvec_keys_lds_kernel, dim3(grid), dim3(256), lds, stream, v, L, D);
  } else {
    hipLaunchKernelGGL(tvec_keys_kernel, dim3(grid_for(total)), dim3(256), 0, stream, v, L, D, pass);
  }
  return hipGetLastError();
}

// exclusive prefix sums of len[0..items) into offs[0..items); *total (device) receives the grand total
hipError_t ring_exclusive_scan(const uint64_t *len, uint64_t *offs, uint64_t items, void *temp, size_t *temp_bytes,
                               hipStream_t stream) {
  return hipcub::DeviceScan::ExclusiveSum(temp, *temp_bytes, len, offs, (int)items, stream);
}

hipError_t launch_groups_insert(const int32_t *gid, const CatCols &cat, uint64_t rows, const CatLayout &L,
                                const CatDevice &D, const CatLayout &Lg, const CatDevice &Dg, int is_key, hipStream_t stream) {
  if (rows == 0 || (!is_key && L.m == 0)) return hipSuccess;
  hipLaunchKernelGGL(groups_insert_kernel, dim3(grid_for(rows)), dim3(256), 0, stream, gid, cat, rows, L, D, Lg, Dg, is_key);
  return hipGetLastError();
}

hipError_t launch_max_i32(const int32_t *v, uint64_t rows, int *out, hipStream_t stream) {
  if (rows == 0) return hipSuccess;
  hipLaunchKernelGGL(max_i32_kernel, dim3(grid_for(rows)), dim3(256), 0, stream, v, rows, out);
  return hipGetLastError();
}

hipError_t launch_groups_accumulate(const int32_t *gid, const NumCols &num, const CatCols &cat, uint64_t rows,
                                    const CatLayout &L, const CatDevice &D, const CatLayout &Lg, const CatDevice &Dg,
                                    int is_key, double *tab, long long dtot, int grid, hipStream_t stream) {
  if (rows == 0) return hipSuccess;
  const size_t lds = (size_t)(GR_THREADS / 64) * 64 * (L.n + L.m + 1) * 4;
  const uint64_t want = (rows + GR_THREADS - 1) / GR_THREADS;
  hipLaunchKernelGGL(groups_accumulate_kernel, dim3((unsigned)std::min<uint64_t>(want, (uint64_t)grid)), dim3(GR_THREADS), lds,
                     stream, gid, num, cat, rows, L, D, Lg, Dg, is_key, tab, dtot);
  return hipGetLastError();
}

hipError_t launch_groups_relayout(const CatLayout &Lo, const CatLayout &Ln, const double *to, double *tn, long long dto,
                                  long long dtn, long long groups, hipStream_t stream) {
  if (groups == 0) return hipSuccess;
  hipLaunchKernelGGL(groups_relayout_kernel, dim3(grid_for((uint64_t)(groups * dto))), dim3(256), 0, stream, Lo, Ln, to, tn, dto, dtn, groups);
  return hipGetLastError();
}

hipError_t launch_groups_combine(double *tab, long long dtot, long long dst, long long src, hipStream_t stream) {
  hipLaunchKernelGGL(groups_combine_kernel, dim3(grid_for((uint64_t)dtot)), dim3(256), 0, stream, tab, dtot, dst, src);
  return hipGetLastError();
}

hipError_t launch_groups_lists(const double *tab, long long dtot, const CatLayout &L, const int32_t *gorder, long long groups,
                               const int32_t *ord, const int32_t *keyof, int family, uint64_t *len, const uint64_t *offs,
                               const cofactor_tvec &out, int mode, hipStream_t stream) {
  const int per_row = family == 0 ? L.m : (family == 1 ? L.n * L.m : tri_i(L.m));
  if (groups == 0 || per_row == 0) return hipSuccess;
  hipLaunchKernelGGL(groups_lists_kernel, dim3(grid_for((uint64_t)(groups * per_row))), dim3(256), 0, stream, tab, dtot, L, gorder,
                     groups, ord, keyof, family, len, offs, out, mode);
  return hipGetLastError();
}

hipError_t launch_groups_dense(const double *tab, long long dtot, int n, int T, const int32_t *gorder, long long groups,
                               const cofactor_tvec &out, hipStream_t stream) {
  if (groups == 0) return hipSuccess;
  hipLaunchKernelGGL(groups_dense_kernel, dim3(grid_for((uint64_t)(groups * (1 + n + T)))), dim3(256), 0, stream, tab, dtot, n, T,
                     gorder, groups, out);
  return hipGetLastError();
}

}  // namespace cofactor
