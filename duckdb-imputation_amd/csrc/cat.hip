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
}

// counts and sums of the key columns in col_mask (their tables: cnt u32 [kc], then s f64 [kc][n])
__global__ __launch_bounds__(CAT_THREADS) void cat_sums_kernel(NumCols num, const unsigned short *__restrict__ codes, uint64_t rows,
                                                               uint64_t stride, CatLayout L, CatDevice D, unsigned col_mask,
                                                               int do_s) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
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
  const uint64_t step = (uint64_t)gridDim.x * CAT_THREADS;
  for (uint64_t r = (uint64_t)blockIdx.x * CAT_THREADS + tid; r < rows; r += step) {
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
        double *row = l_s + s_base[c] + (int)code * n;
#pragma unroll
        for (int k = 0; k < COFACTOR_MAX_NUM; k++)
          if (k < n) unsafeAtomicAdd(&row[k], (double)x[k]);
      }
    }
  }
  __syncthreads();
  for (int c = 0; c < m; c++) {
    if (!((col_mask >> c) & 1u)) continue;
    for (int i = tid; i < L.kc[c]; i += CAT_THREADS)
      if (l_c[c_base[c] + i]) atomicAdd(&D.cnt[L.cnt_off[c] + i], (unsigned long long)l_c[c_base[c] + i]);
    if (do_s)
      for (int i = tid; i < L.kc[c] * n; i += CAT_THREADS)
        if (l_s[s_base[c] + i] != 0.0) unsafeAtomicAdd(&D.s[L.s_off[c] + i], l_s[s_base[c] + i]);
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
  const size_t lds = cat_sums_lds_bytes(L, col_mask, do_s);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)cat_sums_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  const uint64_t need = (rows + CAT_THREADS - 1) / CAT_THREADS;
  if ((uint64_t)grid > need) grid = (int)need;
  hipLaunchKernelGGL(cat_sums_kernel, dim3(grid), dim3(CAT_THREADS), lds, stream, num, codes, rows, stride, L, D, col_mask, do_s ? 1 : 0);
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

hipError_t launch_cat_fold_u32(const unsigned *src, long long cells, unsigned long long *dst, hipStream_t stream) {
  if (cells == 0) return hipSuccess;
  hipLaunchKernelGGL(cat_fold_u32_kernel, dim3((unsigned)std::min<long long>((cells + 255) / 256, 4096)), dim3(256), 0, stream, src, cells, dst);
  return hipGetLastError();
}

}  // namespace cofactor
