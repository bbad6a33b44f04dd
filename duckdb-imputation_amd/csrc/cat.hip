// Categorical part of sum_to_triple_n_m / sum_to_nb_agg_n_m on gfx950: per categorical column
// the per-key row count and per-key sums of every numeric column (lin_cat, quad_num_cat), and per
// column pair the per-(key1,key2) row count (quad_cat).
//
// Replaces the reference's std::map find/insert per row per column and per column pair
// (duckdb_extension/src/triple/sum/sum_no_lift.cpp:157-214, sum_to_nb_agg.cpp:124-145).
//
// Keys are arbitrary int32.  Each column has a device dictionary (open addressing, 64-bit slots
// "valid<<32 | key") that maps a key to a dense code in insertion order; counts, sums and pair
// counts live in dense code-indexed tables.  The sorted key order the reference's std::map gives
// its output is produced at finalize on the host (tables are tiny compared with the scan).
//
// One update is: cat_insert (find unseen keys) -> cat_assign_codes -> [host: grow tables if a
// column outgrew them] -> cat_accumulate.  cat_accumulate keeps private copies of the tables in
// LDS when they fit (ds_add_u32 / ds_add_f64) and flushes them with global atomics at the end;
// otherwise it adds straight into the global tables.
#include "device.hpp"

#include <algorithm>

namespace cofactor {

namespace {

constexpr int CAT_THREADS = 1024;

__device__ __forceinline__ unsigned hash_key(int32_t key, int cap) {
  // cap is a power of two >= 2
  return ((unsigned)key * 0x9E3779B1u) >> (32 - (31 - __builtin_clz(cap)));
}
__device__ __forceinline__ unsigned long long pack_key(int32_t key) {
  return (1ull << 32) | (unsigned long long)(unsigned)key;
}

__device__ __forceinline__ void dict_insert(unsigned long long *slots, int cap, int32_t key, int32_t *flags) {
  const unsigned long long want = pack_key(key);
  unsigned h = hash_key(key, cap);
  for (int probe = 0; probe < cap; probe++) {
    const unsigned long long cur = slots[h];
    if (cur == want) return;
    if (cur == 0ull) {
      const unsigned long long old = atomicCAS(&slots[h], 0ull, want);
      if (old == 0ull || old == want) return;
    }
    h = (h + 1) & (cap - 1);
    // a table that somebody already found full is grown and the batch re-run: stop probing it (a
    // full table of `cap` slots costs `cap` dependent loads per key; 2e7 keys against a table
    // sized for 1e3 took seconds)
    if ((probe & 31) == 31 && __hip_atomic_load(&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  }
  flags[0] = 1;                                   // dictionary full: host grows it and re-runs
}

// Four rows per thread, 16-B loads per column when the column is 16-B aligned.  The dictionaries
// as they stand at launch are copied to LDS (when lds_slots > 0) and a key found there costs one
// LDS probe; only keys missing from that snapshot go to the global table (CAS insert).
__global__ __launch_bounds__(256) void cat_insert_kernel(CatCols cols, uint64_t rows, CatLayout L,
                                                         CatDevice D, int lds_slots) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned long long *l_slot = reinterpret_cast<unsigned long long *>(lds_raw);
  for (int i = threadIdx.x; i < lds_slots; i += blockDim.x) l_slot[i] = D.ht_slot[i];
  if (lds_slots) __syncthreads();
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  constexpr int CH = 10;                          // columns whose loads are issued together
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 4;
  for (uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; r < rows; r += stride) {
    const bool whole = r + 4 <= rows;
    const int cnt = whole ? 4 : (int)(rows - r);
    for (int c0 = 0; c0 < L.m; c0 += CH) {
      i32x4 kv[CH];
#pragma unroll
      for (int j = 0; j < CH; j++) {              // all loads of the chunk first: bytes in flight
        const int c = c0 + j;
        if (c < L.m) {
          const int32_t *col = cols.p[c];
          if (whole && (reinterpret_cast<uintptr_t>(col) & 15) == 0) {
            kv[j] = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(col + r));
          } else {
            i32x4 v = {0, 0, 0, 0};
            if (cnt > 0) v[0] = col[r];
            if (cnt > 1) v[1] = col[r + 1];
            if (cnt > 2) v[2] = col[r + 2];
            if (cnt > 3) v[3] = col[r + 3];
            kv[j] = v;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < CH; j++) {
        const int c = c0 + j;
        if (c < L.m) {
          unsigned long long *slots = D.ht_slot + L.ht_off[c];
          const int cap = L.ht_cap[c];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            if (e >= cnt) break;
            const int32_t key = kv[j][e];
            bool seen = false;
#pragma unroll
            for (int f = 0; f < e; f++) seen = seen || kv[j][f] == key;
            if (!seen && lds_slots) {             // snapshot probe
              const unsigned long long want = pack_key(key);
              unsigned h = hash_key(key, cap);
              for (int probe = 0; probe < cap; probe++) {
                const unsigned long long cur = l_slot[L.ht_off[c] + h];
                if (cur == want) { seen = true; break; }
                if (cur == 0ull) break;
                h = (h + 1) & (cap - 1);
              }
            }
            if (!seen) dict_insert(slots, cap, key, D.flags);
          }
        }
      }
    }
  }
}

// Every occupied slot without a code draws the column's next code.
__global__ __launch_bounds__(256) void cat_assign_kernel(CatLayout L, CatDevice D) {
  const int c = blockIdx.y;
  if (c >= L.m) return;
  const int cap = L.ht_cap[c];
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < cap; s += gridDim.x * blockDim.x) {
    const int g = L.ht_off[c] + s;
    if (D.ht_slot[g] != 0ull && D.ht_code[g] < 0) D.ht_code[g] = atomicAdd(&D.nkeys[c], 1);
  }
}

__global__ __launch_bounds__(256) void cat_rehash_kernel(CatLayout Lo, CatDevice Do, CatLayout Ln,
                                                         CatDevice Dn) {
  const int c = blockIdx.y;
  if (c >= Lo.m) return;
  const int cap_o = Lo.ht_cap[c], cap_n = Ln.ht_cap[c];
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < cap_o; s += gridDim.x * blockDim.x) {
    const unsigned long long v = Do.ht_slot[Lo.ht_off[c] + s];
    if (v == 0ull) continue;
    const int32_t key = (int32_t)(unsigned)(v & 0xffffffffull);
    unsigned h = hash_key(key, cap_n);
    for (int probe = 0; probe < cap_n; probe++) {
      const unsigned long long old = atomicCAS(&Dn.ht_slot[Ln.ht_off[c] + h], 0ull, v);
      if (old == 0ull) { Dn.ht_code[Ln.ht_off[c] + h] = Do.ht_code[Lo.ht_off[c] + s]; break; }
      h = (h + 1) & (cap_n - 1);
    }
  }
}

__global__ __launch_bounds__(256) void cat_relayout_kernel(CatLayout Lo, CatDevice Do, CatLayout Ln,
                                                           CatDevice Dn) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  for (int i = tid; i < Lo.n_cnt; i += nth) {     // counts: same code, new column offset
    int c = 0;
    while (c + 1 < Lo.m && i >= Lo.cnt_off[c + 1]) c++;
    Dn.cnt[Ln.cnt_off[c] + (i - Lo.cnt_off[c])] = Do.cnt[i];
  }
  if (Lo.kind == 0) {
    for (int i = tid; i < Lo.n_s; i += nth) {     // sums: [code][k] keeps its shape
      int c = 0;
      while (c + 1 < Lo.m && i >= Lo.s_off[c + 1]) c++;
      Dn.s[Ln.s_off[c] + (i - Lo.s_off[c])] = Do.s[i];
    }
    const int npairs = Lo.m * (Lo.m + 1) / 2;
    for (int i = tid; i < Lo.n_p; i += nth) {     // pairs: row stride kc[c2] changes
      int q = 0;
      while (q + 1 < npairs && i >= Lo.p_off[q + 1]) q++;
      int c1 = 0, rem = q;
      while (rem >= Lo.m - c1) { rem -= Lo.m - c1; c1++; }
      const int c2 = c1 + rem;
      if (pair_is_sparse(Ln, q)) continue;         // (the host moves such a table into its sparse store)
      const int local = i - Lo.p_off[q];
      const int code1 = local / Lo.kc[c2], code2 = local % Lo.kc[c2];
      Dn.p[Ln.p_off[q] + code1 * Ln.kc[c2] + code2] = Do.p[i];
    }
  }
}

// Looks `key` up in a dictionary that is known to contain it.
template <typename SlotPtr, typename CodePtr>
__device__ __forceinline__ int lookup_code(SlotPtr slots, CodePtr codes, int cap, int32_t key) {
  const unsigned long long want = pack_key(key);
  unsigned h = hash_key(key, cap);
  for (int probe = 0; probe < cap; probe++) {
    if (slots[h] == want) return codes[h];
    h = (h + 1) & (cap - 1);
  }
  return -1;
}

// One launch accumulates the tables named by `P`: the counts, the per-key sums, and the pair tables
// whose bit is set in P.pair_mask.  The host splits an update into passes whose tables fit LDS
// (api.cpp: plan_cat_passes); tables too big for LDS are updated with global atomics.
// MT > 0: the number of categorical columns is the compile-time constant MT (straight-line
// atomics, no per-column branches); MT == 0: generic variant that reads m from the layout.
template <bool LDS_TABLES, int KIND, int MT>
__global__ __launch_bounds__(CAT_THREADS) void cat_accumulate_kernel(NumCols num, CatCols cat,
                                                                     uint64_t rows, CatLayout L,
                                                                     CatDevice D, CatPass P,
                                                                     const uint8_t *__restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  // LDS carve: [dictionary slots (8 B) | sums (8 B) | dictionary codes | counts | pairs (4 B each)]
  const int d_slots = P.dict_lds ? L.n_slots : 0;
  const int t_s = (LDS_TABLES && KIND == 0 && P.do_s) ? L.n_s : 0;
  const int t_cnt = (LDS_TABLES && P.do_cnt) ? L.n_cnt : 0;
  const int t_p = (LDS_TABLES && KIND == 0) ? P.p_cells : 0;
  unsigned long long *l_slot = reinterpret_cast<unsigned long long *>(lds_raw);
  double *l_s = reinterpret_cast<double *>(l_slot + d_slots);
  int32_t *l_code = reinterpret_cast<int32_t *>(l_s + t_s);
  unsigned *l_cnt = reinterpret_cast<unsigned *>(l_code + d_slots);
  unsigned *l_p = l_cnt + t_cnt;
  unsigned short *l_direct = reinterpret_cast<unsigned short *>(l_p + t_p);   // [column][key 0..255 | other] -> code
  constexpr int MC = MT > 0 ? MT : COFACTOR_MAX_CAT;       // unroll bound
  const int m = MT > 0 ? MT : L.m;
  unsigned char *l_far = reinterpret_cast<unsigned char *>(l_direct + m * CAT_DIRECT_STRIDE);   // [column]: a key outside 0..255

  const int tid = threadIdx.x;
  for (int i = tid; i < d_slots; i += CAT_THREADS) { l_slot[i] = D.ht_slot[i]; l_code[i] = D.ht_code[i]; }
  for (int i = tid; i < t_cnt; i += CAT_THREADS) l_cnt[i] = 0u;
  for (int i = tid; i < t_s; i += CAT_THREADS) l_s[i] = 0.0;
  for (int i = tid; i < t_p; i += CAT_THREADS) l_p[i] = 0u;
  if (P.dict_lds) {
    for (int i = tid; i < m * CAT_DIRECT_STRIDE; i += CAT_THREADS) l_direct[i] = 0xFFFFu;
    if (tid < 32) l_far[tid] = 0;
  }
  __syncthreads();
  if (P.dict_lds) {
    for (int c = 0; c < m; c++)
      for (int i = tid; i < L.ht_cap[c]; i += CAT_THREADS) {
        const unsigned long long sv = l_slot[L.ht_off[c] + i];
        const int32_t cdv = l_code[L.ht_off[c] + i];
        if (sv != 0ull && cdv >= 0) {
          const unsigned key = (unsigned)(sv & 0xFFFFFFFFull);
          if (key < (unsigned)CAT_DIRECT_KEYS && cdv < 0xFFFF) l_direct[c * CAT_DIRECT_STRIDE + key] = (unsigned short)cdv;
          else l_far[c] = 1;
        }
      }
    __syncthreads();
  }
  // dictionary through generic pointers: LDS copy when it fits, HBM otherwise
  const unsigned long long *dict_slot = P.dict_lds ? l_slot : D.ht_slot;
  const int32_t *dict_code = P.dict_lds ? l_code : D.ht_code;

#ifdef COFACTOR_DEV_ABLATE   // timing experiments only (results are wrong when a bit is set)
  const int ablate = D.flags[2];
#else
  constexpr int ablate = 0;
#endif
  const int n = L.n;
  const bool do_s = KIND == 0 && P.do_s && !(ablate & 2);
  const uint64_t stride = (uint64_t)gridDim.x * CAT_THREADS;
  for (uint64_t r = (uint64_t)blockIdx.x * CAT_THREADS + tid; r < rows; r += stride) {
    if (mask && mask[r] == 0) continue;                     // masked update: this row is filtered out
    int32_t key[MC];
#pragma unroll
    for (int c = 0; c < MC; c++)
      if (c < m && ((P.col_mask >> c) & 1u)) key[c] = cat.p[c][r];   // all loads first, then the probes
    float xk = (do_s && n > 0) ? num.p[0][r] : 0.f;
    int code[MC];
    bool known = true;
#pragma unroll
    for (int c = 0; c < MC; c++) {
      if (c < m && ((P.col_mask >> c) & 1u)) {
        if (P.dict_lds && !l_far[c]) {                      // keys 0..255: one table read
          const unsigned e = l_direct[c * CAT_DIRECT_STRIDE + min((unsigned)key[c], (unsigned)CAT_DIRECT_KEYS)];
          code[c] = e == 0xFFFFu ? -1 : (int)e;
        } else {
          code[c] = lookup_code(dict_slot + L.ht_off[c], dict_code + L.ht_off[c], L.ht_cap[c], key[c]);
        }
        known = known && code[c] >= 0 && code[c] < L.kc[c];
      }
    }
    if (!known) { D.flags[1] = 1; continue; }               // never index a table with a bad code
    if (P.do_cnt && !(ablate & 1)) {
#pragma unroll
      for (int c = 0; c < MC; c++) {
        if (c < m) {
          if (LDS_TABLES) atomicAdd(&l_cnt[L.cnt_off[c] + code[c]], 1u);
          else atomicAdd(&D.cnt[L.cnt_off[c] + code[c]], 1ull);
        }
      }
    }
    if (KIND == 0) {
      if (!(ablate & 4)) {
        int q = 0;
#pragma unroll
        for (int c1 = 0; c1 < MC; c1++) {
#pragma unroll
          for (int c2 = c1; c2 < MC; c2++) {
            if (c2 < m) {
              if ((P.pair_mask[q >> 5] >> (q & 31)) & 1u) {
                const int idx = L.p_off[q] + code[c1] * L.kc[c2] + code[c2];
                if (LDS_TABLES) atomicAdd(&l_p[idx - P.p_base], 1u);
                else atomicAdd(&D.p[idx], 1ull);
              }
              q++;
            }
          }
        }
      }
      if (do_s) {
        int sidx[MC];
#pragma unroll
        for (int c = 0; c < MC; c++)
          if (c < m) sidx[c] = L.s_off[c] + code[c] * n;
        for (int k = 0; k < n; k++) {
          const double x = (double)xk;
          if (k + 1 < n) xk = num.p[k + 1][r];              // next column's value flies under the adds
#pragma unroll
          for (int c = 0; c < MC; c++) {
            if (c < m) {
              if (LDS_TABLES) unsafeAtomicAdd(&l_s[sidx[c] + k], x);
              else unsafeAtomicAdd(&D.s[sidx[c] + k], x);
            }
          }
        }
      }
    }
  }

  if (LDS_TABLES) {
    __syncthreads();
    for (int i = tid; i < t_cnt; i += CAT_THREADS)
      if (l_cnt[i]) atomicAdd(&D.cnt[i], (unsigned long long)l_cnt[i]);
    for (int i = tid; i < t_s; i += CAT_THREADS)
      if (l_s[i] != 0.0) unsafeAtomicAdd(&D.s[i], l_s[i]);
    for (int i = tid; i < t_p; i += CAT_THREADS)
      if (l_p[i]) atomicAdd(&D.p[P.p_base + i], (unsigned long long)l_p[i]);
  }
}

template <bool LT, int K, int MT>
hipError_t launch_acc(const NumCols &num, const CatCols &cat, uint64_t rows, const CatLayout &L,
                      const CatDevice &D, const CatPass &P, int grid, size_t lds, const uint8_t *mask,
                      hipStream_t stream) {
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)cat_accumulate_kernel<LT, K, MT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((cat_accumulate_kernel<LT, K, MT>), dim3(grid), dim3(CAT_THREADS), lds, stream,
                     num, cat, rows, L, D, P, mask);
  return hipGetLastError();
}

template <bool LT, int K>
hipError_t launch_acc_m(int m, const NumCols &num, const CatCols &cat, uint64_t rows,
                        const CatLayout &L, const CatDevice &D, const CatPass &P, int grid, size_t lds,
                        const uint8_t *mask, hipStream_t stream) {
  switch (m) {
#define CASE(M) case M: return launch_acc<LT, K, M>(num, cat, rows, L, D, P, grid, lds, mask, stream);
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10)
    CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18) CASE(19) CASE(20)
#undef CASE
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

size_t cat_pass_lds_bytes(const CatLayout &L, const CatPass &P, bool lds_tables) {
  size_t b = P.dict_lds ? (size_t)L.n_slots * (8 + 4) + cat_direct_lds_bytes(L.m) : 0;
  if (lds_tables) {
    if (P.do_cnt) b += (size_t)L.n_cnt * 4;
    if (L.kind == 0 && P.do_s) b += (size_t)L.n_s * 8;
    if (L.kind == 0) b += (size_t)P.p_cells * 4;
  }
  return b;
}

hipError_t launch_cat_insert(const CatCols &cols, uint64_t rows, const CatLayout &L,
                             const CatDevice &D, hipStream_t stream) {
  if (rows == 0 || L.m == 0) return hipSuccess;
  uint64_t blocks = (rows + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  const int lds_slots = (size_t)L.n_slots * 8 <= 48 * 1024 ? L.n_slots : 0;
  hipLaunchKernelGGL(cat_insert_kernel, dim3((unsigned)blocks), dim3(256), (size_t)lds_slots * 8, stream,
                     cols, rows, L, D, lds_slots);
  return hipGetLastError();
}

hipError_t launch_cat_assign_codes(const CatLayout &L, const CatDevice &D, hipStream_t stream) {
  if (L.m == 0) return hipSuccess;
  int maxcap = 0;
  for (int c = 0; c < L.m; c++) maxcap = L.ht_cap[c] > maxcap ? L.ht_cap[c] : maxcap;
  int bx = (maxcap + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(cat_assign_kernel, dim3(bx, L.m), dim3(256), 0, stream, L, D);
  return hipGetLastError();
}

hipError_t launch_cat_rehash(const CatLayout &Lold, const CatDevice &Dold, const CatLayout &Lnew,
                             const CatDevice &Dnew, hipStream_t stream) {
  if (Lold.m == 0) return hipSuccess;
  int maxcap = 0;
  for (int c = 0; c < Lold.m; c++) maxcap = Lold.ht_cap[c] > maxcap ? Lold.ht_cap[c] : maxcap;
  int bx = (maxcap + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(cat_rehash_kernel, dim3(bx, Lold.m), dim3(256), 0, stream, Lold, Dold, Lnew, Dnew);
  return hipGetLastError();
}

hipError_t launch_cat_relayout(const CatLayout &Lold, const CatDevice &Dold, const CatLayout &Lnew,
                               const CatDevice &Dnew, hipStream_t stream) {
  if (Lold.m == 0) return hipSuccess;
  hipLaunchKernelGGL(cat_relayout_kernel, dim3(512), dim3(256), 0, stream, Lold, Dold, Lnew, Dnew);
  return hipGetLastError();
}

hipError_t launch_cat_accumulate(const NumCols &num, const CatCols &cat, uint64_t rows,
                                 const CatLayout &L, const CatDevice &D, const CatPass &P,
                                 bool lds_tables, int grid, hipStream_t stream, hipEvent_t ev0,
                                 hipEvent_t ev1, const uint8_t *mask) {
  if (rows == 0 || L.m == 0) return hipSuccess;
  const uint64_t need = (rows + CAT_THREADS - 1) / CAT_THREADS;
  if ((uint64_t)grid > need) grid = (int)need;
  const size_t lds = cat_pass_lds_bytes(L, P, lds_tables);
  if (ev0) { hipError_t e = hipEventRecord(ev0, stream); if (e != hipSuccess) return e; }
  hipError_t le;
  if (lds_tables)
    le = L.kind == 0 ? launch_acc_m<true, 0>(L.m, num, cat, rows, L, D, P, grid, lds, mask, stream)
                     : launch_acc_m<true, 1>(L.m, num, cat, rows, L, D, P, grid, lds, mask, stream);
  else
    le = L.kind == 0 ? launch_acc_m<false, 0>(L.m, num, cat, rows, L, D, P, grid, lds, mask, stream)
                     : launch_acc_m<false, 1>(L.m, num, cat, rows, L, D, P, grid, lds, mask, stream);
  if (le != hipSuccess) return le;
  if (ev1) return hipEventRecord(ev1, stream);
  return hipSuccess;
}

// ---- dictionary-aligned table seam of the multi-GPU path -----------------
// Moves every accumulated value from the old code of its key to the new one (remap[cnt_off_old[c] +
// old code] = new code, -1 for unused codes).  remap is injective per column: plain stores into
// the freshly zeroed new tables.
__global__ __launch_bounds__(256) void cat_remap_kernel(CatLayout Lo, CatDevice Do, CatLayout Ln,
                                                        CatDevice Dn, const int32_t *__restrict__ remap) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
  for (long long i = tid; i < Lo.n_cnt; i += nth) {
    int c = 0;
    while (c + 1 < Lo.m && i >= Lo.cnt_off[c + 1]) c++;
    const int nc = remap[i];
    if (nc >= 0) Dn.cnt[Ln.cnt_off[c] + nc] = Do.cnt[i];
  }
  if (Lo.kind != 0) return;
  for (long long i = tid; i < Lo.n_s; i += nth) {
    int c = 0;
    while (c + 1 < Lo.m && i >= Lo.s_off[c + 1]) c++;
    const int local = (int)(i - Lo.s_off[c]);
    const int code = local / Lo.n, k = local % Lo.n;
    const int nc = remap[Lo.cnt_off[c] + code];
    if (nc >= 0) Dn.s[Ln.s_off[c] + (long long)nc * Ln.n + k] = Do.s[i];
  }
  const int npairs = Lo.m * (Lo.m + 1) / 2;
  for (long long i = tid; i < Lo.n_p; i += nth) {
    const unsigned long long v = Do.p[i];
    if (!v) continue;
    int q = 0;
    while (q + 1 < npairs && i >= Lo.p_off[q + 1]) q++;
    if (pair_is_sparse(Ln, q)) continue;           // (moved into its sorted store by the caller)
    int c1 = 0, rem = q;
    while (rem >= Lo.m - c1) { rem -= Lo.m - c1; c1++; }
    const int c2 = c1 + rem;
    const int local = (int)(i - Lo.p_off[q]);
    const int n1 = remap[Lo.cnt_off[c1] + local / Lo.kc[c2]], n2 = remap[Lo.cnt_off[c2] + local % Lo.kc[c2]];
    if (n1 >= 0 && n2 >= 0) Dn.p[Ln.p_off[q] + (long long)n1 * Ln.kc[c2] + n2] = v;
  }
}

// [cnt | s | p] as ONE array of doubles (counts are exact integers below 2^53) and back.
// add != 0: the values are added to the tables instead of replacing them.
__global__ __launch_bounds__(256) void cat_tables_export_kernel(CatLayout L, CatDevice D, double *__restrict__ out) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
  const long long total = (long long)L.n_cnt + L.n_s + L.n_p;
  for (long long i = tid; i < total; i += nth) {
    double v;
    if (i < L.n_cnt) v = (double)D.cnt[i];
    else if (i < (long long)L.n_cnt + L.n_s) v = D.s[i - L.n_cnt];
    else v = (double)D.p[i - L.n_cnt - L.n_s];
    out[i] = v;
  }
}
__global__ __launch_bounds__(256) void cat_tables_import_kernel(CatLayout L, CatDevice D, const double *__restrict__ in,
                                                                int add) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
  const long long total = (long long)L.n_cnt + L.n_s + L.n_p;
  for (long long i = tid; i < total; i += nth) {
    const double v = in[i];
    if (i < L.n_cnt) D.cnt[i] = (add ? D.cnt[i] : 0ull) + (unsigned long long)(v + 0.5);
    else if (i < (long long)L.n_cnt + L.n_s) D.s[i - L.n_cnt] = (add ? D.s[i - L.n_cnt] : 0.0) + v;
    else D.p[i - L.n_cnt - L.n_s] = (add ? D.p[i - L.n_cnt - L.n_s] : 0ull) + (unsigned long long)(v + 0.5);
  }
}

hipError_t launch_cat_remap(const CatLayout &Lold, const CatDevice &Dold, const CatLayout &Lnew,
                            const CatDevice &Dnew, const int32_t *remap, hipStream_t stream) {
  if (Lold.m == 0) return hipSuccess;
  const long long cells = std::max<long long>(Lold.n_p, std::max(Lold.n_s, Lold.n_cnt));
  const int grid = (int)std::min<long long>(4096, (cells + 255) / 256);
  hipLaunchKernelGGL(cat_remap_kernel, dim3(grid), dim3(256), 0, stream, Lold, Dold, Lnew, Dnew, remap);
  return hipGetLastError();
}

hipError_t launch_cat_tables_export(const CatLayout &L, const CatDevice &D, double *out, hipStream_t stream) {
  const long long total = (long long)L.n_cnt + L.n_s + L.n_p;
  if (total == 0) return hipSuccess;
  const int grid = (int)std::min<long long>(4096, (total + 255) / 256);
  hipLaunchKernelGGL(cat_tables_export_kernel, dim3(grid), dim3(256), 0, stream, L, D, out);
  return hipGetLastError();
}

hipError_t launch_cat_tables_import(const CatLayout &L, const CatDevice &D, const double *in, bool add,
                                    hipStream_t stream) {
  const long long total = (long long)L.n_cnt + L.n_s + L.n_p;
  if (total == 0) return hipSuccess;
  const int grid = (int)std::min<long long>(4096, (total + 255) / 256);
  hipLaunchKernelGGL(cat_tables_import_kernel, dim3(grid), dim3(256), 0, stream, L, D, in, add ? 1 : 0);
  return hipGetLastError();
}

// ---- multi-pass generic path with a code cache ---------------------------------------------------------
// When the tables of an update do not fit LDS together, the keys are translated ONCE into 16-bit
// codes ([column][row], 0xFFFF = row dropped / key unknown) and every later pass reads those: 2 bytes
// per key instead of 4, no dictionary probes, four rows per thread.
//   cat_codes_kernel   keys -> codes
//   cat_sums_kernel    key counts + per-key sums of a subset of the key columns (LDS tables)
//   cat_pairs_reg_kernel  pair tables of a run of column pairs (LDS tables)
//   cat_pair_hbm_kernel   ONE pair table too big for LDS as u32 cells in HBM: it is the only table
//                         written in that launch, so it stays in L2 / MALL (1 M cells = 4 MB)
//                         instead of 55 tables thrashing HBM
constexpr unsigned short CODE_NONE = 0xFFFFu;

// miss_slot: which flag a key missing from its dictionary raises — 1 (sticky error: the dictionary
// pass has run, this cannot happen) or 3 (optimistic run without a dictionary pass: the host then
// runs the pass and redoes the batch; nothing has been accumulated yet).
template <bool LDS_DICT>
__global__ __launch_bounds__(256) void cat_codes_kernel(CatCols cat, uint64_t rows, uint64_t stride, CatLayout L, CatDevice D,
                                                        const uint8_t *__restrict__ mask, unsigned short *__restrict__ codes,
                                                        int miss_slot) {
  // four rows per thread, one column at a time: 16-B key loads, 8-B code stores; the dictionaries
  // are probed in an LDS copy when they fit (LDS_DICT), else in HBM / L2
  extern __shared__ __attribute__((aligned(16))) unsigned char codes_lds[];
  unsigned long long *l_slot = reinterpret_cast<unsigned long long *>(codes_lds);
  int32_t *l_code = reinterpret_cast<int32_t *>(l_slot + (LDS_DICT ? L.n_slots : 0));
  if (LDS_DICT) {
    for (int i = threadIdx.x; i < L.n_slots; i += 256) { l_slot[i] = D.ht_slot[i]; l_code[i] = D.ht_code[i]; }
    __syncthreads();
  }
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const uint64_t nq = (rows + 3) / 4;
  for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = 4 * q;
    const int cnt = (int)min<uint64_t>(4, rows - r);
    unsigned keep = 0xF;
    if (mask) { keep = 0; for (int e = 0; e < cnt; e++) keep |= (mask[r + e] != 0) << e; }
    for (int c = 0; c < L.m; c++) {
      const int32_t *col = cat.p[c];
      i32x4 kv = {0, 0, 0, 0};
      if (cnt == 4 && (reinterpret_cast<uintptr_t>(col) & 15) == 0) kv = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(col + r));
      else for (int e = 0; e < cnt; e++) kv[e] = col[r + e];
      unsigned short out[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        out[e] = CODE_NONE;
        if (e < cnt && ((keep >> e) & 1)) {
          const int code = LDS_DICT ? cat_lookup_code(l_slot + L.ht_off[c], l_code + L.ht_off[c], L.ht_cap[c], kv[e])
                                    : cat_lookup_code(D.ht_slot + L.ht_off[c], D.ht_code + L.ht_off[c], L.ht_cap[c], kv[e]);
          if (code < 0 || code >= L.kc[c]) D.flags[miss_slot] = 1;
          else if (code < 0xFFFF) out[e] = (unsigned short)code;   // (beyond: a wide column, never read from the cache)
        }
      }
      // stride is a multiple of 4: the store is 8-byte aligned
      *reinterpret_cast<uint2 *>(codes + (uint64_t)c * stride + r) = make_uint2(out[0] | ((unsigned)out[1] << 16), out[2] | ((unsigned)out[3] << 16));
    }
  }
  // behind the last group of four, up to the end of its 64-row tile: CODE_NONE (the stride covers it)
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    const uint64_t pad = (rows + 3) / 4 * 4 + threadIdx.x, end = (rows + 63) / 64 * 64;
    if (pad < end)
      for (int c = 0; c < L.m; c++) codes[(uint64_t)c * stride + pad] = CODE_NONE;
  }
}

// counts and sums of the key columns in col_mask (their tables: cnt u32 [kc], then s f64 [kc][n]).
// subs.n > 0: ONE launch for several column subsets whose tables do not fit LDS together — workgroup
// b serves subset b % subs.n over the rows of row group b / subs.n, so the subs.n workgroups that
// read the same rows run side by side and the numeric columns come from HBM once (the others hit
// L2 / the Infinity Cache) instead of once per subset: 10 launches of 4.2 GB each at K = 1000.
struct SumSubsets { int n; unsigned mask[COFACTOR_MAX_CAT]; };
__global__ __launch_bounds__(CAT_THREADS) void cat_sums_kernel(NumCols num, const unsigned short *__restrict__ codes, uint64_t rows,
                                                               uint64_t stride, CatLayout L, CatDevice D, unsigned col_mask,
                                                               int do_s, SumSubsets subs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned bid = blockIdx.x, nblk = gridDim.x;
  if (subs.n > 0) { col_mask = subs.mask[bid % subs.n]; bid /= subs.n; nblk /= subs.n; }
  // LDS carve: sums first (8-byte aligned), then counts
  __shared__ int s_base[COFACTOR_MAX_CAT], c_base[COFACTOR_MAX_CAT];
  __shared__ int tot_s, tot_c;
  const int n = L.n, m = L.m, tid = threadIdx.x;
  if (tid == 0) {
    int so = 0, co = 0;
    for (int c = 0; c < m; c++)
      if ((col_mask >> c) & 1u) { s_base[c] = so; c_base[c] = co; so += do_s ? L.kc[c] * n : 0; co += L.kc[c]; }
    tot_s = so; tot_c = co;
  }
  __syncthreads();
  double *l_s = reinterpret_cast<double *>(lds_raw);
  unsigned *l_c = reinterpret_cast<unsigned *>(l_s + tot_s);
  for (int i = tid; i < tot_s; i += CAT_THREADS) l_s[i] = 0.0;
  for (int i = tid; i < tot_c; i += CAT_THREADS) l_c[i] = 0u;
  __syncthreads();
  const uint64_t step = (uint64_t)nblk * CAT_THREADS;
  for (uint64_t r = (uint64_t)bid * CAT_THREADS + tid; r < rows; r += step) {
    float x[COFACTOR_MAX_NUM];
    if (do_s)
#pragma unroll
      for (int k = 0; k < COFACTOR_MAX_NUM; k++)
        if (k < n) x[k] = num.p[k][r];
    for (int c = 0; c < m; c++) {
      if (!((col_mask >> c) & 1u)) continue;
      const unsigned code = codes[(uint64_t)c * stride + r];
      if (code == CODE_NONE) continue;
      atomicAdd(&l_c[c_base[c] + code], 1u);
      if (do_s) {
        // in LDS the sums of a column lie [numeric column][code]: the 64 lanes of one ds_add_f64 differ
        // in the code only, and consecutive codes are consecutive banks ([code][numeric column], the
        // order of the global table, puts codes 20 words apart at n = 10: 16 bank pairs for 64 lanes)
        const int kc = L.kc[c];
        double *col = l_s + s_base[c] + (int)code;
#pragma unroll
        for (int k = 0; k < COFACTOR_MAX_NUM; k++)
          if (k < n) unsafeAtomicAdd(&col[k * kc], (double)x[k]);
      }
    }
  }
  __syncthreads();
  for (int c = 0; c < m; c++) {
    if (!((col_mask >> c) & 1u)) continue;
    for (int i = tid; i < L.kc[c]; i += CAT_THREADS)
      if (l_c[c_base[c] + i]) atomicAdd(&D.cnt[L.cnt_off[c] + i], (unsigned long long)l_c[c_base[c] + i]);
    if (do_s)
      for (int i = tid; i < L.kc[c] * n; i += CAT_THREADS) {          // i = code * n + k in the global table
        const int code = i / n, k = i - code * n;
        const double v = l_s[s_base[c] + k * L.kc[c] + code];
        if (v != 0.0) unsafeAtomicAdd(&D.s[L.s_off[c] + i], v);
      }
  }
}

// ONE pair table too big for LDS: u32 cells in `gtab` (zeroed by the caller, folded into D.p by
// cat_fold_u32_kernel); P.pair_mask names the pair.
__global__ __launch_bounds__(CAT_THREADS) void cat_pair_hbm_kernel(const unsigned short *__restrict__ codes, uint64_t rows,
                                                                   uint64_t stride, CatLayout L, CatPass P,
                                                                   unsigned *__restrict__ gtab) {
  const int m = L.m, tid = threadIdx.x;
  int c1 = 0, c2 = 0;
  {
    int q = 0;
    for (int a = 0; a < m; a++)
      for (int b = a; b < m; b++, q++)
        if ((P.pair_mask[q >> 5] >> (q & 31)) & 1u) { c1 = a; c2 = b; }
  }
  const int kc2 = L.kc[c2];
  const uint64_t nq = (rows + 3) / 4, step = (uint64_t)gridDim.x * CAT_THREADS;
  for (uint64_t qd = (uint64_t)blockIdx.x * CAT_THREADS + tid; qd < nq; qd += step) {
    const uint64_t r = 4 * qd;
    const uint2 a = *reinterpret_cast<const uint2 *>(codes + (uint64_t)c1 * stride + r);
    const uint2 b = c2 == c1 ? a : *reinterpret_cast<const uint2 *>(codes + (uint64_t)c2 * stride + r);
    const unsigned ca[4] = {a.x & 0xFFFFu, a.x >> 16, a.y & 0xFFFFu, a.y >> 16};
    const unsigned cb[4] = {b.x & 0xFFFFu, b.x >> 16, b.y & 0xFFFFu, b.y >> 16};
#pragma unroll
    for (int e = 0; e < 4; e++)
      if (ca[e] != CODE_NONE && cb[e] != CODE_NONE) atomicAdd(&gtab[ca[e] * kc2 + cb[e]], 1u);   // (rows past the end hold CODE_NONE)
  }
}

// Pair tables that fit LDS together (several pairs per launch: P.pair_mask; u32 cells in LDS
// covering D.p cells [P.p_base, P.p_base + P.p_cells), added to D.p at the end), with the 4 rows'
// codes of ALL key columns in registers: re-loading column c2's codes for every pair (c1, c2), as
// the first version did, is ~105 loads of 8 bytes per 4 rows and launch at 20 columns — 10 GB per
// 5e7-row launch out of L2 / HBM, more than the table scan itself.  MC = m rounded up to a
// multiple of 4 (compile-time: register arrays need static indices); the pair loops are fully
// unrolled, a pair outside this launch's mask costs one scalar bit test.
template <int MC>
__global__ __launch_bounds__(CAT_THREADS) void cat_pairs_reg_kernel(const unsigned short *__restrict__ codes, uint64_t rows,
                                                                    uint64_t stride, CatLayout L, CatDevice D, CatPass P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned *l_p = reinterpret_cast<unsigned *>(lds_raw);
  const int m = L.m, tid = threadIdx.x;
  for (int i = tid; i < P.p_cells; i += CAT_THREADS) l_p[i] = 0u;
  __syncthreads();
  const uint64_t nq = (rows + 3) / 4, step = (uint64_t)gridDim.x * CAT_THREADS;
  for (uint64_t qd = (uint64_t)blockIdx.x * CAT_THREADS + tid; qd < nq; qd += step) {
    const uint64_t r = 4 * qd;
    uint2 cd[MC];
#pragma unroll
    for (int c = 0; c < MC; c++)
      cd[c] = c < m ? *reinterpret_cast<const uint2 *>(codes + (uint64_t)c * stride + r) : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
#pragma unroll
    for (int c1 = 0; c1 < MC; c1++) {
#pragma unroll
      for (int c2 = c1; c2 < MC; c2++) {
        if (c2 >= m) continue;
        const int q = c1 * m - c1 * (c1 - 1) / 2 + (c2 - c1);
        if (!((P.pair_mask[q >> 5] >> (q & 31)) & 1u)) continue;
        const int off = L.p_off[q] - P.p_base, kc2 = L.kc[c2];
        const unsigned ca[4] = {cd[c1].x & 0xFFFFu, cd[c1].x >> 16, cd[c1].y & 0xFFFFu, cd[c1].y >> 16};
        const unsigned cb[4] = {cd[c2].x & 0xFFFFu, cd[c2].x >> 16, cd[c2].y & 0xFFFFu, cd[c2].y >> 16};
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (ca[e] != CODE_NONE && cb[e] != CODE_NONE) atomicAdd(&l_p[off + ca[e] * kc2 + cb[e]], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < P.p_cells; i += CAT_THREADS)
    if (l_p[i]) atomicAdd(&D.p[P.p_base + i], (unsigned long long)l_p[i]);
}

__global__ __launch_bounds__(256) void cat_fold_u32_kernel(const unsigned *__restrict__ src, long long cells,
                                                           unsigned long long *__restrict__ dst) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (long long)gridDim.x * blockDim.x)
    if (src[i]) dst[i] += src[i];
}

hipError_t launch_cat_codes(const CatCols &cat, uint64_t rows, uint64_t stride, const CatLayout &L, const CatDevice &D,
                            const uint8_t *mask, unsigned short *codes, hipStream_t stream, bool optimistic) {
  const int miss_slot = optimistic ? 3 : 1;
  if (rows == 0 || L.m == 0) return hipSuccess;
  const uint64_t nq = (rows + 3) / 4;
  const int grid = (int)std::min<uint64_t>((nq + 255) / 256, 8192);
  const size_t dict = (size_t)L.n_slots * 12;
  if (dict <= 48 * 1024)
    hipLaunchKernelGGL((cat_codes_kernel<true>), dim3(grid), dim3(256), dict, stream, cat, rows, stride, L, D, mask, codes, miss_slot);
  else
    hipLaunchKernelGGL((cat_codes_kernel<false>), dim3(grid), dim3(256), 0, stream, cat, rows, stride, L, D, mask, codes, miss_slot);
  return hipGetLastError();
}

size_t cat_sums_lds_bytes(const CatLayout &L, unsigned col_mask, bool do_s) {
  size_t b = 0;
  for (int c = 0; c < L.m; c++)
    if ((col_mask >> c) & 1u) b += (size_t)L.kc[c] * 4 + (do_s ? (size_t)L.kc[c] * L.n * 8 : 0);
  return b;
}

hipError_t launch_cat_sums(const NumCols &num, const unsigned short *codes, uint64_t rows, uint64_t stride,
                           const CatLayout &L, const CatDevice &D, unsigned col_mask, int grid, hipStream_t stream) {
  if (rows == 0 || col_mask == 0) return hipSuccess;
  const bool do_s = L.kind == 0 && L.n > 0;
  // 17 .. 64 codes per column: counts and sums on the matrix cores (catsums.hip); COFACTOR_NO_SUMS_MFMA=1: the LDS atomics
  static const bool no_mfma = [] { const char *v = getenv("COFACTOR_NO_SUMS_MFMA"); return v && *v == '1'; }();
  if (do_s && !no_mfma && stride % 64 == 0 && cat_sums_mfma_applicable(L, col_mask, rows))
    return launch_cat_sums_mfma(num, codes, rows, stride, L, D, col_mask, grid, stream);
  const size_t lds = cat_sums_lds_bytes(L, col_mask, do_s);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)cat_sums_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  const uint64_t need = (rows + CAT_THREADS - 1) / CAT_THREADS;
  if ((uint64_t)grid > need) grid = (int)need;
  hipLaunchKernelGGL(cat_sums_kernel, dim3(grid), dim3(CAT_THREADS), lds, stream, num, codes, rows, stride, L, D, col_mask, do_s ? 1 : 0,
                     SumSubsets{});
  return hipGetLastError();
}

hipError_t launch_cat_sums_subsets(const NumCols &num, const unsigned short *codes, uint64_t rows, uint64_t stride,
                                   const CatLayout &L, const CatDevice &D, const unsigned *masks, int nsub, int cus,
                                   hipStream_t stream) {
  if (rows == 0 || nsub <= 0) return hipSuccess;
  if (nsub > COFACTOR_MAX_CAT) return hipErrorInvalidValue;
  const bool do_s = L.kind == 0 && L.n > 0;
  SumSubsets subs{};
  subs.n = nsub;
  size_t lds = 0;
  for (int i = 0; i < nsub; i++) { subs.mask[i] = masks[i]; lds = std::max(lds, cat_sums_lds_bytes(L, masks[i], do_s)); }
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)cat_sums_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  // all workgroups resident at once (one per CU when the tables take most of its LDS), so that the
  // subsets of a row group really run side by side
  const int per_cu = std::max(1, (int)std::min<size_t>(4, (150 * 1024) / std::max<size_t>(lds, 1)));
  int groups = std::max(1, cus * per_cu / nsub);
  const uint64_t need = (rows + CAT_THREADS - 1) / CAT_THREADS;
  if ((uint64_t)groups > need) groups = (int)need;
  hipLaunchKernelGGL(cat_sums_kernel, dim3(groups * nsub), dim3(CAT_THREADS), lds, stream, num, codes, rows, stride, L, D, 0u,
                     do_s ? 1 : 0, subs);
  return hipGetLastError();
}

hipError_t launch_cat_pairs(const unsigned short *codes, uint64_t rows, uint64_t stride, const CatLayout &L, const CatDevice &D,
                            const CatPass &P, unsigned *gtab, int grid, hipStream_t stream) {
  if (rows == 0) return hipSuccess;
  const uint64_t need = ((rows + 3) / 4 + CAT_THREADS - 1) / CAT_THREADS;
  if ((uint64_t)grid > need) grid = (int)need;
  if (gtab) {
    hipLaunchKernelGGL(cat_pair_hbm_kernel, dim3(grid), dim3(CAT_THREADS), 0, stream, codes, rows, stride, L, P, gtab);
  } else {
    const size_t lds = (size_t)P.p_cells * 4;
    hipError_t e = hipErrorInvalidValue;
    switch ((L.m + 3) / 4) {
#define CASE(Q) case Q: \
        e = lds > 48 * 1024 ? hipFuncSetAttribute((const void *)cat_pairs_reg_kernel<4 * Q>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) : hipSuccess; \
        if (e == hipSuccess) hipLaunchKernelGGL((cat_pairs_reg_kernel<4 * Q>), dim3(grid), dim3(CAT_THREADS), lds, stream, codes, rows, stride, L, D, P); \
        break;
      CASE(1) CASE(2) CASE(3) CASE(4) CASE(5)
#undef CASE
      default: break;
    }
    if (e != hipSuccess) return e;
  }
  return hipGetLastError();
}

// ---- pair tables too big for LDS: rows BINNED by the high bits of code 1 (round 3) ------------------------
// cat_pair_hbm_kernel does one global atomic per row and pair into a 4 MB table: 2.7e10 atomics/s, the
// chip's rate for scattered 4-byte atomics, 1.86 ms per pair and 5e7 rows at K = 1000 (72 % of that
// step).  LDS atomics are ~100x faster but a 1024 x 1024 table does not fit.  A SLICE of it does:
// `sb` rows of code 1 x all kc2 codes of column 2 (32 x 1024 u32 = 128 KB).  So, per column c1 that
// has such pairs:
//   cat_bin_count_kernel    histogram of (code1 >> shift) over the piece (all c1 in one launch)
//   cat_bin_scan_kernel     bin offsets = exclusive scan; cursors start at the offsets
//   cat_bin_scatter_kernel  rows regrouped bin by bin: the low bits of code 1 and the codes of every
//                           partner column c2, staged through LDS so that each bin's run of a
//                           workgroup's chunk leaves as contiguous 2-byte stores
//   cat_pair_bin_kernel     one workgroup per (pair, bin[, split]): its slice in LDS, ds_add_u32 per
//                           row, the slice added to the u64 table at the end (plain adds: a cell
//                           belongs to one bin)
// Traffic: the partner codes are read and written once per c1 (2 B each) and read once per pair.
constexpr int BIN_CHUNK = 8192;                 // rows of a scatter workgroup's chunk (32 per thread, 8 quads)
constexpr int BIN_MAX = 1024;                   // bins per column at most (LDS histograms)

__global__ __launch_bounds__(256) void cat_bin_count_kernel(const unsigned short *__restrict__ codes, uint64_t rows,
                                                            uint64_t stride, const BinPlan *__restrict__ planp,
                                                            unsigned *__restrict__ hist) {
  const BinPlan &plan = *planp;
  __shared__ unsigned l_hist[BIN_MAX];
  for (int j = 0; j < plan.ncols; j++) {
    const int nb = plan.nb[j], shift = plan.shift[j];
    const unsigned short *col = codes + (uint64_t)plan.col[j] * stride;
    for (int i = threadIdx.x; i < nb; i += 256) l_hist[i] = 0u;
    __syncthreads();
    const uint64_t nq = (rows + 3) / 4;
    for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (uint64_t)gridDim.x * 256) {
      const uint2 a = *reinterpret_cast<const uint2 *>(col + 4 * q);     // (rows past the end hold CODE_NONE)
      const unsigned c[4] = {a.x & 0xFFFFu, a.x >> 16, a.y & 0xFFFFu, a.y >> 16};
#pragma unroll
      for (int e = 0; e < 4; e++)
        if (c[e] != CODE_NONE) atomicAdd(&l_hist[c[e] >> shift], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += 256)
      if (l_hist[i]) atomicAdd(&hist[j * BIN_MAX + i], l_hist[i]);
    __syncthreads();
  }
}

// off[j][0..nb] = exclusive scan of hist[j][..]; cursor[j][b] = off[j][b]; hist is cleared for the next piece
__global__ __launch_bounds__(64) void cat_bin_scan_kernel(const BinPlan *__restrict__ planp, unsigned *__restrict__ hist,
                                                          unsigned *__restrict__ off, unsigned *__restrict__ cursor) {
  const BinPlan &plan = *planp;
  const int j = blockIdx.x;
  if (threadIdx.x != 0 || j >= plan.ncols) return;
  unsigned run = 0;
  for (int b = 0; b < plan.nb[j]; b++) {
    off[j * (BIN_MAX + 1) + b] = run;
    cursor[j * BIN_MAX + b] = run;
    run += hist[j * BIN_MAX + b];
    hist[j * BIN_MAX + b] = 0u;
  }
  off[j * (BIN_MAX + 1) + plan.nb[j]] = run;
}

// out[0][pos] = code1 & (sb - 1), out[1 + i][pos] = code of partner column i, pos = the row's place in
// its bin (bin-major).  One chunk of BIN_CHUNK rows per workgroup and step.
__global__ __launch_bounds__(256) void cat_bin_scatter_kernel(const unsigned short *__restrict__ codes, uint64_t rows,
                                                              uint64_t stride, const BinPlan *__restrict__ planp, int j,
                                                              unsigned *__restrict__ cursor,
                                                              unsigned short *__restrict__ out, uint64_t out_stride) {
  const BinPlan &plan = *planp;
  __shared__ unsigned l_cnt[BIN_MAX], l_off[BIN_MAX], l_base[BIN_MAX];
  __shared__ unsigned l_gpos[BIN_CHUNK];
  __shared__ unsigned short l_stage[BIN_CHUNK];
  const int nb = plan.nb[j], shift = plan.shift[j], tid = threadIdx.x;
  const unsigned lowmask = (1u << shift) - 1u;
  const unsigned short *col1 = codes + (uint64_t)plan.col[j] * stride;
  unsigned *cur = cursor + j * BIN_MAX;
  constexpr int PER = BIN_CHUNK / 256;
  const uint64_t nchunks = (rows + BIN_CHUNK - 1) / BIN_CHUNK;
  for (uint64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const uint64_t r0 = ch * BIN_CHUNK;
    for (int i = tid; i < nb; i += 256) l_cnt[i] = 0u;
    __syncthreads();
    // a thread's rows: quads r0 + 4 (q 256 + tid) .. + 3 (8-byte loads; rows past the end hold CODE_NONE)
    unsigned code1[PER], rank[PER];
#pragma unroll
    for (int q = 0; q < PER / 4; q++) {
      const uint64_t r = r0 + 4 * ((uint64_t)q * 256 + tid);
      uint2 v = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
      if (r < stride) v = *reinterpret_cast<const uint2 *>(col1 + r);
      code1[4 * q] = v.x & 0xFFFFu; code1[4 * q + 1] = v.x >> 16; code1[4 * q + 2] = v.y & 0xFFFFu; code1[4 * q + 3] = v.y >> 16;
    }
#pragma unroll
    for (int e = 0; e < PER; e++) rank[e] = code1[e] != CODE_NONE ? atomicAdd(&l_cnt[code1[e] >> shift], 1u) : 0u;
    __syncthreads();
    if (tid < 64) {                                   // one wave: exclusive scan of the bin counts + global reservation
      unsigned carry = 0;
      for (int b0 = 0; b0 < nb; b0 += 64) {
        const int b = b0 + tid;
        const unsigned v = b < nb ? l_cnt[b] : 0u;
        unsigned incl = v;
        for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d, 64); if (tid >= d) incl += o; }
        if (b < nb) {
          l_off[b] = carry + incl - v;
          l_base[b] = v ? atomicAdd(&cur[b], v) : 0u;
        }
        carry += __shfl(incl, 63, 64);
      }
    }
    __syncthreads();
    unsigned slot[PER];
#pragma unroll
    for (int e = 0; e < PER; e++) {
      slot[e] = 0xFFFFFFFFu;
      if (code1[e] != CODE_NONE) {
        const unsigned b = code1[e] >> shift;
        slot[e] = l_off[b] + rank[e];
        l_gpos[slot[e]] = l_base[b] + rank[e];
      }
    }
    unsigned total = 0;
    {
      const int last = nb - 1;
      __syncthreads();
      total = l_off[last] + l_cnt[last];
    }
    for (int k = 0; k <= plan.npart[j]; k++) {        // column 0: low bits of code 1; then the partner columns
      const unsigned short *src = k == 0 ? nullptr : codes + (uint64_t)plan.part[j][k - 1] * stride;
#pragma unroll
      for (int q = 0; q < PER / 4; q++) {
        const uint64_t r = r0 + 4 * ((uint64_t)q * 256 + tid);
        uint2 v = make_uint2(0u, 0u);
        if (k != 0 && r < stride) v = *reinterpret_cast<const uint2 *>(src + r);
        const unsigned val[4] = {v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16};
#pragma unroll
        for (int j4 = 0; j4 < 4; j4++) {
          const int e = 4 * q + j4;
          if (slot[e] != 0xFFFFFFFFu) l_stage[slot[e]] = k == 0 ? (unsigned short)(code1[e] & lowmask) : (unsigned short)val[j4];
        }
      }
      __syncthreads();
      unsigned short *dst = out + (uint64_t)k * out_stride;
      // two staged elements per thread and step: one 4-byte store when they are neighbours in the
      // same bin run at an even position (most are: a chunk's run per bin is ~ BIN_CHUNK / nb long)
      for (unsigned i = 2 * tid; i < total; i += 512) {
        const unsigned g0 = l_gpos[i];
        if (i + 1 < total) {
          const unsigned g1 = l_gpos[i + 1];
          if (g1 == g0 + 1 && !(g0 & 1u)) {
            *reinterpret_cast<unsigned *>(dst + g0) = (unsigned)l_stage[i] | ((unsigned)l_stage[i + 1] << 16);
          } else {
            dst[g0] = l_stage[i];
            dst[g1] = l_stage[i + 1];
          }
        } else {
          dst[g0] = l_stage[i];
        }
      }
      __syncthreads();
    }
  }
}

// One workgroup per (partner k, bin b, split s) of column plan.col[j]: the rows of the bin's s-th part,
// slice [sb][kc2] of the pair table in LDS.
__global__ __launch_bounds__(1024) void cat_pair_bin_kernel(const unsigned short *__restrict__ binned, uint64_t out_stride,
                                                            const BinPlan *__restrict__ planp, int j,
                                                            const unsigned *__restrict__ off, int splits,
                                                            unsigned long long *__restrict__ p) {
  const BinPlan &plan = *planp;
  extern __shared__ __attribute__((aligned(16))) unsigned char bin_lds[];
  unsigned *slice = reinterpret_cast<unsigned *>(bin_lds);
  const int nb = plan.nb[j], sb = 1 << plan.shift[j];
  const int s = blockIdx.x % splits, b = (blockIdx.x / splits) % nb, k = blockIdx.x / (splits * nb);
  const int kc2 = plan.part_kc[j][k], kbits = 31 - __builtin_clz(kc2);
  const int cells = sb * kc2;
  for (int i = threadIdx.x; i < cells; i += 1024) slice[i] = 0u;
  __syncthreads();
  const unsigned *o = off + j * (BIN_MAX + 1);
  const uint64_t lo = o[b], hi = o[b + 1], len = hi - lo;
  const uint64_t a = lo + len * s / splits, e = lo + len * (s + 1) / splits;
  const unsigned short *c1 = binned, *c2 = binned + (uint64_t)(1 + k) * out_stride;
  // the body in quads of rows from an 8-byte aligned start, head and tail row by row
  uint64_t r = a;
  const uint64_t body0 = min(e, (a + 3) / 4 * 4), body1 = body0 + (e > body0 ? (e - body0) / 4 * 4 : 0);
  for (uint64_t i = r + threadIdx.x; i < body0; i += 1024) atomicAdd(&slice[((unsigned)c1[i] << kbits) + c2[i]], 1u);
  for (uint64_t q = body0 / 4 + threadIdx.x; q < body1 / 4; q += 1024) {
    const uint2 x = *reinterpret_cast<const uint2 *>(c1 + 4 * q), y = *reinterpret_cast<const uint2 *>(c2 + 4 * q);
    atomicAdd(&slice[((x.x & 0xFFFFu) << kbits) + (y.x & 0xFFFFu)], 1u);
    atomicAdd(&slice[((x.x >> 16) << kbits) + (y.x >> 16)], 1u);
    atomicAdd(&slice[((x.y & 0xFFFFu) << kbits) + (y.y & 0xFFFFu)], 1u);
    atomicAdd(&slice[((x.y >> 16) << kbits) + (y.y >> 16)], 1u);
  }
  for (uint64_t i = body1 + threadIdx.x; i < e; i += 1024) atomicAdd(&slice[((unsigned)c1[i] << kbits) + c2[i]], 1u);
  __syncthreads();
  unsigned long long *tab = p + plan.part_poff[j][k] + (uint64_t)b * (uint64_t)cells;   // rows [b sb, (b+1) sb) of the table
  for (int i = threadIdx.x; i < cells; i += 1024) {
    const unsigned v = slice[i];
    if (v) {
      if (splits > 1) atomicAdd(&tab[i], (unsigned long long)v);
      else tab[i] += v;
    }
  }
}

size_t bin_scratch_words() { return (size_t)COFACTOR_MAX_CAT * (3 * BIN_MAX + 1); }

hipError_t launch_cat_binned_pairs(const unsigned short *codes, uint64_t rows, uint64_t stride, const BinPlan &plan,
                                   const BinPlan *d_plan, unsigned *scratch, unsigned short *binned, uint64_t out_stride,
                                   int cus, unsigned long long *p, hipStream_t stream) {
  if (rows == 0 || plan.ncols == 0) return hipSuccess;
  unsigned *hist = scratch, *off = hist + COFACTOR_MAX_CAT * BIN_MAX, *cursor = off + COFACTOR_MAX_CAT * (BIN_MAX + 1);
  const int grid = (int)std::min<uint64_t>((uint64_t)cus * 8, (rows / 4 + 255) / 256);
  hipLaunchKernelGGL(cat_bin_count_kernel, dim3(std::max(1, grid)), dim3(256), 0, stream, codes, rows, stride, d_plan, hist);
  hipLaunchKernelGGL(cat_bin_scan_kernel, dim3(plan.ncols), dim3(64), 0, stream, d_plan, hist, off, cursor);
  hipError_t e = hipGetLastError();
  for (int j = 0; j < plan.ncols && e == hipSuccess; j++) {
    const int sgrid = (int)std::min<uint64_t>((uint64_t)cus * 4, (rows + BIN_CHUNK - 1) / BIN_CHUNK);
    hipLaunchKernelGGL(cat_bin_scatter_kernel, dim3(std::max(1, sgrid)), dim3(256), 0, stream, codes, rows, stride, d_plan, j,
                       cursor, binned, out_stride);
    int max_cells = 0;
    for (int k = 0; k < plan.npart[j]; k++) max_cells = std::max(max_cells, (1 << plan.shift[j]) * plan.part_kc[j][k]);
    const size_t lds = (size_t)max_cells * 4;
    int splits = 1;
    while (plan.npart[j] * plan.nb[j] * splits < 2 * cus && splits < 8) splits *= 2;
    if ((e = hipFuncSetAttribute((const void *)cat_pair_bin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) break;
    hipLaunchKernelGGL(cat_pair_bin_kernel, dim3(plan.npart[j] * plan.nb[j] * splits), dim3(1024), lds, stream, binned, out_stride,
                       d_plan, j, off, splits, p);
    e = hipGetLastError();
  }
  return e;
}

// ---- finalize: the quad_cat lists written on the device (SumStateFinalize, sum_state.cpp:440-461) ---------
// One workgroup per table row in ascending key order (row t = code `row_code[t]` of pair `row_pair[t]`):
//   pairlist_count_kernel   non-zero cells of the row among the live codes of column 2
//   pairlist_fill_kernel    (key1, key2, count) triples as doubles at rowbase[t], in ascending key2
// The host only turns 55 000 row counts into offsets; the 5.5e7 cells of a K = 1000 state never
// cross PCIe as a 440 MB table to be re-encoded by host threads, they arrive as the finished list.
__global__ __launch_bounds__(256) void pairlist_count_kernel(const unsigned long long *__restrict__ p,
                                                             const int *__restrict__ row_pair, const int *__restrict__ row_code,
                                                             const PairListInfo *__restrict__ info,
                                                             const int *__restrict__ order_flat, unsigned *__restrict__ rowcnt) {
  const int t = blockIdx.x;
  const PairListInfo pi = info[row_pair[t]];
  const unsigned long long *row = p + pi.p_off + (long long)row_code[t] * pi.kc2;
  const int *o2 = order_flat + pi.o2_off;
  unsigned nz = 0;
  for (int j = threadIdx.x; j < pi.o2_len; j += 256) nz += row[o2[j]] != 0ull;
  for (int off = 32; off > 0; off >>= 1) nz += __shfl_down(nz, off, 64);
  __shared__ unsigned part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = nz;
  __syncthreads();
  if (threadIdx.x == 0) rowcnt[t] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void pairlist_fill_kernel(const unsigned long long *__restrict__ p,
                                                            const int *__restrict__ row_pair, const int *__restrict__ row_code,
                                                            const PairListInfo *__restrict__ info,
                                                            const int *__restrict__ order_flat, const int *__restrict__ key_flat,
                                                            const unsigned long long *__restrict__ rowbase,
                                                            double *__restrict__ out) {
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const PairListInfo pi = info[row_pair[t]];
  const int code1 = row_code[t];
  const unsigned long long *row = p + pi.p_off + (long long)code1 * pi.kc2;
  const int *o2 = order_flat + pi.o2_off;
  const double key1 = (double)key_flat[pi.k1_off + code1];
  double *w = out + rowbase[t];
  __shared__ unsigned wsum[4];
  __shared__ unsigned carry;
  if (tid == 0) carry = 0u;
  __syncthreads();
  for (int j0 = 0; j0 < pi.o2_len; j0 += 256) {
    const int j = j0 + tid;
    int code2 = 0;
    unsigned long long v = 0ull;
    if (j < pi.o2_len) { code2 = o2[j]; v = row[code2]; }
    const unsigned flag = v != 0ull;
    unsigned incl = flag;
    for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned before = carry;
    for (int k = 0; k < wv; k++) before += wsum[k];
    if (flag) {
      double *e = w + 3ull * (before + incl - 1u);
      e[0] = key1; e[1] = (double)key_flat[pi.k2_off + code2]; e[2] = (double)v;
    }
    __syncthreads();
    if (tid == 0) carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
}

hipError_t launch_pairlist_count(const unsigned long long *p, const int *row_pair, const int *row_code, int rows,
                                 const PairListInfo *info, const int *order_flat, unsigned *rowcnt, hipStream_t stream) {
  if (rows == 0) return hipSuccess;
  hipLaunchKernelGGL(pairlist_count_kernel, dim3(rows), dim3(256), 0, stream, p, row_pair, row_code, info, order_flat, rowcnt);
  return hipGetLastError();
}
hipError_t launch_pairlist_fill(const unsigned long long *p, const int *row_pair, const int *row_code, int rows,
                                const PairListInfo *info, const int *order_flat, const int *key_flat,
                                const unsigned long long *rowbase, double *out, hipStream_t stream) {
  if (rows == 0) return hipSuccess;
  hipLaunchKernelGGL(pairlist_fill_kernel, dim3(rows), dim3(256), 0, stream, p, row_pair, row_code, info, order_flat, key_flat,
                     rowbase, out);
  return hipGetLastError();
}

hipError_t launch_cat_fold_u32(const unsigned *src, long long cells, unsigned long long *dst, hipStream_t stream) {
  if (cells == 0) return hipSuccess;
  hipLaunchKernelGGL(cat_fold_u32_kernel, dim3((unsigned)std::min<long long>((cells + 255) / 256, 4096)), dim3(256), 0, stream, src, cells, dst);
  return hipGetLastError();
}

}  // namespace cofactor
