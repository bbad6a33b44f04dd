// C ABI of libcofactor_hip.so (include/cofactor_hip.h): contexts, aggregate states, the update /
// combine / finalize life cycle of the reference's aggregates, and the scalar ring ops.
// There is no CPU fallback for the aggregate: without a GPU cofactor_ctx_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "device.hpp"
#include "ml.hpp"
#include "state.hpp"
#include "triple.hpp"

using namespace cofactor;

namespace cofactor {
namespace detail {

thread_local std::string g_err;

cofactor_status fail(cofactor_status st, const std::string &msg) {
  g_err = msg;
  return st;
}
cofactor_status hip_fail(hipError_t e, const char *what) {
  return fail(COFACTOR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

long env_long(const char *name, long dflt) {
  const char *v = std::getenv(name);
  if (!v || !*v) return dflt;
  char *end = nullptr;
  long x = std::strtol(v, &end, 10);
  return (end && *end == 0 && x > 0) ? x : dflt;
}

int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace detail
}  // namespace cofactor

using namespace cofactor::detail;

namespace cofactor {
namespace detail {

void cat_free(CatDevice &D) {
  (void)hipFree(D.ht_slot); (void)hipFree(D.ht_code); (void)hipFree(D.nkeys); (void)hipFree(D.flags);
  (void)hipFree(D.cnt); (void)hipFree(D.s); (void)hipFree(D.p);
  D = CatDevice{};
}

// A pair table with more than this many cells is kept as a sparse list (sparse.hpp); so are the
// largest remaining ones while the dense total would pass 2^30 cells.  COFACTOR_SPARSE_CELLS
// overrides the threshold (tests).
static uint64_t sparse_threshold() { return (uint64_t)env_long("COFACTOR_SPARSE_CELLS", 1l << 26); }

// Fills offsets / totals of L from its n, m, kind, ht_cap[], kc[].  False if a table would
// overflow the 31-bit indices the kernels use.  allow_sparse: oversized pair tables are marked
// sparse (no dense cells) instead of failing — the aggregate path only.
bool cat_finish_layout(CatLayout &L, bool allow_sparse) {
  uint64_t slots = 0, cnt = 0, s = 0, p = 0;
  for (int c = 0; c < L.m; c++) {
    L.ht_off[c] = (int)slots; slots += (uint64_t)L.ht_cap[c];
    L.cnt_off[c] = (int)cnt;  cnt += (uint64_t)L.kc[c];
    L.s_off[c] = (int)s;      s += (uint64_t)L.kc[c] * (uint64_t)L.n;
    if (slots > (1ull << 30) || cnt > (1ull << 30) || s > (1ull << 30)) return false;
  }
  for (auto &w : L.sparse_mask) w = 0u;
  const int npairs = L.m * (L.m + 1) / 2;
  std::vector<uint64_t> cells(npairs, 0);
  {
    int q = 0;
    for (int c1 = 0; c1 < L.m; c1++)
      for (int c2 = c1; c2 < L.m; c2++, q++) cells[q] = (uint64_t)L.kc[c1] * (uint64_t)L.kc[c2];
  }
  if (L.kind == 0 && allow_sparse) {
    uint64_t dense = 0;
    for (int q = 0; q < npairs; q++) {
      if (cells[q] > sparse_threshold()) L.sparse_mask[q >> 5] |= 1u << (q & 31);
      else dense += cells[q];
    }
    while (dense > (1ull << 30)) {                // still too much: the largest dense table goes sparse
      int big = -1;
      for (int q = 0; q < npairs; q++)
        if (!pair_is_sparse(L, q) && (big < 0 || cells[q] > cells[big])) big = q;
      if (big < 0) break;
      L.sparse_mask[big >> 5] |= 1u << (big & 31);
      dense -= cells[big];
    }
  }
  for (int q = 0; q < npairs; q++) {
    L.p_off[q] = (int)p;
    if (!pair_is_sparse(L, q)) p += cells[q];
    if (p > (1ull << 30)) return false;
  }
  L.n_slots = (int)slots; L.n_cnt = (int)cnt;
  L.n_s = L.kind == 0 ? (int)s : 0;
  L.n_p = L.kind == 0 ? (int)p : 0;
  return true;
}

cofactor_status cat_alloc(const CatLayout &L, CatDevice &D, bool fresh_counters, hipStream_t st) {
  D = CatDevice{};
  HIP_TRY(hipMalloc((void **)&D.ht_slot, sizeof(unsigned long long) * std::max(1, L.n_slots)));
  HIP_TRY(hipMalloc((void **)&D.ht_code, sizeof(int32_t) * std::max(1, L.n_slots)));
  HIP_TRY(hipMalloc((void **)&D.cnt, sizeof(unsigned long long) * std::max(1, L.n_cnt)));
  HIP_TRY(hipMalloc((void **)&D.s, sizeof(double) * std::max(1, L.n_s)));
  HIP_TRY(hipMalloc((void **)&D.p, sizeof(unsigned long long) * std::max(1, L.n_p)));
  HIP_TRY(hipMemsetAsync(D.ht_slot, 0, sizeof(unsigned long long) * std::max(1, L.n_slots), st));
  HIP_TRY(hipMemsetAsync(D.ht_code, 0xFF, sizeof(int32_t) * std::max(1, L.n_slots), st));
  HIP_TRY(hipMemsetAsync(D.cnt, 0, sizeof(unsigned long long) * std::max(1, L.n_cnt), st));
  HIP_TRY(hipMemsetAsync(D.s, 0, sizeof(double) * std::max(1, L.n_s), st));
  HIP_TRY(hipMemsetAsync(D.p, 0, sizeof(unsigned long long) * std::max(1, L.n_p), st));
  if (fresh_counters) {
    HIP_TRY(hipMalloc((void **)&D.nkeys, sizeof(int32_t) * COFACTOR_MAX_CAT));
    HIP_TRY(hipMalloc((void **)&D.flags, sizeof(int32_t) * 4));
    HIP_TRY(hipMemsetAsync(D.nkeys, 0, sizeof(int32_t) * COFACTOR_MAX_CAT, st));
    HIP_TRY(hipMemsetAsync(D.flags, 0, sizeof(int32_t) * 4, st));
  }
  return COFACTOR_OK;
}

// Replaces the aggregate's dictionary and/or tables by ones with layout Lnew, carrying over
// every key, code and accumulated value.
cofactor_status cat_regrow(cofactor_agg *a, const CatLayout &Lnew) {
  hipStream_t st = a->ctx->stream;
  CatDevice Dn;
  cofactor_status s = cat_alloc(Lnew, Dn, false, st);
  if (s != COFACTOR_OK) return s;
  Dn.nkeys = a->D.nkeys;
  Dn.flags = a->D.flags;
  hipError_t e = launch_cat_rehash(a->L, a->D, Lnew, Dn, st);
  if (e == hipSuccess) e = launch_cat_relayout(a->L, a->D, Lnew, Dn, st);
  // a pair table that is dense in the old layout and sparse in the new one moves into its store
  if (e == hipSuccess && a->kind == COFACTOR_TRIPLE && any_sparse_pair(Lnew)) {
    const CatLayout &Lo = a->L;
    int32_t *key_of = nullptr;
    a->sparse.resize(tri(a->m));
    int q = 0;
    for (int c1 = 0; c1 < a->m && e == hipSuccess; c1++)
      for (int c2 = c1; c2 < a->m && e == hipSuccess; c2++, q++) {
        if (!pair_is_sparse(Lnew, q) || pair_is_sparse(Lo, q) || !a->dev_dirty) continue;
        if (!key_of) {
          e = hipMalloc((void **)&key_of, sizeof(int32_t) * std::max(1, Lo.n_cnt));
          for (int c = 0; c < a->m && e == hipSuccess; c++)
            e = launch_key_of_code(a->D.ht_slot + Lo.ht_off[c], a->D.ht_code + Lo.ht_off[c], Lo.ht_cap[c], Lo.kc[c],
                                   key_of + Lo.cnt_off[c], st);
        }
        if (e == hipSuccess)
          e = sparse_add_dense(a->ctx->sparse_sc, a->sparse[q], a->D.p + Lo.p_off[q], Lo.kc[c1], Lo.kc[c2],
                               key_of + Lo.cnt_off[c1], key_of + Lo.cnt_off[c2], st);
      }
    if (key_of) { (void)hipStreamSynchronize(st); (void)hipFree(key_of); }
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    Dn.nkeys = nullptr; Dn.flags = nullptr;
    cat_free(Dn);
    return hip_fail(e, "cat_regrow");
  }
  CatDevice old = a->D;
  old.nkeys = nullptr; old.flags = nullptr;     // kept
  cat_free(old);
  a->D = Dn;
  a->L = Lnew;
  return COFACTOR_OK;
}

cofactor_status cat_prepare(cofactor_agg *a) {
  if (a->cat_ready || a->m == 0) return COFACTOR_OK;
  CatLayout &L = a->L;
  L = CatLayout{};
  L.n = a->n; L.m = a->m; L.kind = a->kind;
  for (int c = 0; c < a->m; c++) { L.ht_cap[c] = 64; L.kc[c] = 16; }
  if (!cat_finish_layout(L, true)) return fail(COFACTOR_ERR_UNSUPPORTED, "categorical layout overflow");
  cofactor_status s = cat_alloc(L, a->D, true, a->ctx->stream);
  if (s != COFACTOR_OK) return s;
  a->cat_ready = true;
  return COFACTOR_OK;
}

// Dictionary maintenance for one batch: find unseen keys, give them codes, grow the dictionaries
// and code-indexed tables when a column outgrew them.  Leaves the per-column key counts in
// a->nkeys_host.
cofactor_status cat_dictionaries(cofactor_agg *a, const CatCols &cat, uint64_t rows) {
  return cat_dictionaries_with(a, [&]() { return launch_cat_insert(cat, rows, a->L, a->D, a->ctx->stream); });
}

// the same with any kernel that inserts a batch's keys into a->D's dictionaries (a->L's geometry)
cofactor_status cat_dictionaries_with(cofactor_agg *a, const std::function<hipError_t()> &insert) {
  cofactor_status s = cat_prepare(a);
  if (s != COFACTOR_OK) return s;
  hipStream_t st = a->ctx->stream;
  int32_t counters[COFACTOR_MAX_CAT + 4];
  for (int attempt = 0;; attempt++) {
    HIP_TRY(hipMemsetAsync(a->D.flags, 0, sizeof(int32_t), st));   // [0] only; [1] is sticky
    HIP_TRY(insert());
    HIP_TRY(hipMemcpyAsync(counters + COFACTOR_MAX_CAT, a->D.flags, sizeof(int32_t) * 4,
                           hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (counters[COFACTOR_MAX_CAT + 1])             // sticky: set by an accumulate kernel of an earlier update
      return fail(COFACTOR_ERR_INTERNAL, "a row of an earlier update met a key missing from its dictionary");
    if (counters[COFACTOR_MAX_CAT] == 0) break;
    if (attempt > 24) return fail(COFACTOR_ERR_UNSUPPORTED, "dictionary growth did not converge");
    CatLayout Ln = a->L;                          // a dictionary ran full: quadruple all of them
    for (int c = 0; c < a->m; c++) {
      if (Ln.ht_cap[c] >= (1 << 28)) return fail(COFACTOR_ERR_UNSUPPORTED, "categorical column has too many distinct keys");
      Ln.ht_cap[c] *= 4;
    }
    if (!cat_finish_layout(Ln, true)) return fail(COFACTOR_ERR_UNSUPPORTED, "categorical dictionaries too large");
    s = cat_regrow(a, Ln);
    if (s != COFACTOR_OK) return s;
  }
  HIP_TRY(launch_cat_assign_codes(a->L, a->D, st));
  HIP_TRY(hipMemcpyAsync(counters, a->D.nkeys, sizeof(int32_t) * COFACTOR_MAX_CAT,
                         hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  bool grow = false;
  CatLayout Ln = a->L;
  for (int c = 0; c < a->m; c++) {
    if (a->nkeys_host[c] != counters[c]) a->dict_sig = 0;   // a new key: no longer the aligned dictionary
    a->nkeys_host[c] = counters[c];
    if (counters[c] > Ln.kc[c]) { Ln.kc[c] = next_pow2(counters[c]); grow = true; }
    while (counters[c] * 2 > Ln.ht_cap[c]) { Ln.ht_cap[c] *= 2; grow = true; }  // load factor <= 1/2
  }
  if (grow) {
    if (!cat_finish_layout(Ln, true))
      return fail(COFACTOR_ERR_UNSUPPORTED, "categorical column has too many distinct keys for its count / sum tables");
    s = cat_regrow(a, Ln);
    if (s != COFACTOR_OK) return s;
  }
  return COFACTOR_OK;
}

// Splits the categorical tables of one update into launches whose tables fit the LDS budget:
// counts + sums first, then runs of consecutive pair tables; a pair table that does not fit on its
// own is updated in HBM (global atomics), all such pairs in one last launch.
void plan_cat_passes(const CatLayout &L, size_t lds_budget, std::vector<CatPass> &lds_passes,
                     CatPass &hbm_pass, bool &hbm_needed) {
  lds_passes.clear();
  hbm_needed = false;
  hbm_pass = CatPass{};
  const size_t dict = (size_t)L.n_slots * 12 + cat_direct_lds_bytes(L.m);   // dictionaries + byte-key tables
  const bool dict_lds = dict <= lds_budget / 2;
  const size_t budget = dict_lds ? lds_budget - dict : lds_budget;
  hbm_pass.dict_lds = dict_lds && dict <= 48 * 1024;
  const int npairs = L.kind == 0 ? L.m * (L.m + 1) / 2 : 0;
  std::vector<int> c1_of(npairs), c2_of(npairs);
  for (int c1 = 0, q = 0; c1 < L.m; c1++)
    for (int c2 = c1; c2 < L.m && q < npairs; c2++, q++) { c1_of[q] = c1; c2_of[q] = c2; }
  auto fresh = [&]() { CatPass p{}; p.dict_lds = dict_lds; p.p_base = 0; p.p_cells = 0; return p; };
  // counts (+ sums)
  const size_t base_bytes = (size_t)L.n_cnt * 4 + (L.kind == 0 ? (size_t)L.n_s * 8 : 0);
  CatPass cur = fresh();
  size_t used = 0;
  bool cur_open = false;
  if (base_bytes <= budget) {
    cur.do_cnt = 1; cur.do_s = L.kind == 0; cur.col_mask = (L.m >= 32) ? 0xFFFFFFFFu : ((1u << L.m) - 1u);
    used = base_bytes; cur_open = true;
  } else {
    hbm_pass.do_cnt = 1; hbm_pass.do_s = L.kind == 0;
    hbm_pass.col_mask = (1u << L.m) - 1u;
    hbm_needed = true;
  }
  for (int q = 0; q < npairs; q++) {
    if (pair_is_sparse(L, q)) continue;           // (kept as a sorted list: cat_accumulate's last step)
    const size_t cells = (size_t)L.kc[c1_of[q]] * (size_t)L.kc[c2_of[q]];
    const size_t bytes = cells * 4;
    if (bytes > budget) {                         // too big for LDS on its own
      hbm_pass.pair_mask[q >> 5] |= 1u << (q & 31);
      hbm_pass.col_mask |= (1u << c1_of[q]) | (1u << c2_of[q]);
      hbm_needed = true;
      if (cur_open && cur.p_cells > 0) { lds_passes.push_back(cur); cur = fresh(); used = 0; cur_open = false; }
      continue;
    }
    if (cur_open && used + bytes > budget) { lds_passes.push_back(cur); cur = fresh(); used = 0; cur_open = false; }
    if (!cur_open) { cur = fresh(); used = 0; cur_open = true; }
    if (cur.p_cells == 0) cur.p_base = L.p_off[q];
    cur.pair_mask[q >> 5] |= 1u << (q & 31);
    cur.col_mask |= (1u << c1_of[q]) | (1u << c2_of[q]);
    cur.p_cells += (int)cells;
    used += bytes;
  }
  if (cur_open) lds_passes.push_back(cur);
}

// grows a context scratch buffer (synchronises: kernels of earlier calls may still read the old one)
template <typename T>
cofactor_status scratch_reserve(cofactor_ctx *ctx, T *&buf, size_t &have, size_t need) {
  if (need <= have) return COFACTOR_OK;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  (void)hipFree(buf);
  buf = nullptr;
  have = 0;
  HIP_TRY(hipMalloc((void **)&buf, need));
  have = need;
  return COFACTOR_OK;
}

// grows the per-workgroup pair slabs of the one-pass kernels
cofactor_status ensure_pair_slabs(cofactor_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->pair_slab_bytes) return COFACTOR_OK;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  (void)hipFree(ctx->pair_slabs);
  ctx->pair_slabs = nullptr;
  ctx->pair_slab_bytes = 0;
  HIP_TRY(hipMalloc((void **)&ctx->pair_slabs, bytes));
  ctx->pair_slab_bytes = bytes;
  return COFACTOR_OK;
}

// The categorical tables of shapes the one-pass kernels do not take in one launch.
//  * Everything in one launch of the LDS-atomic kernel when all tables fit LDS together and the
//    one-hot MFMA kernel cannot help.
//  * Otherwise per piece of <= 2^27 rows:
//      - with <= 16 keys in every column (triple kind), the whole 256-row tiles of aligned columns
//        get their key counts and per-key sums from fused2_kernel SUB-LAUNCHES over groups of
//        <= 10 key columns x <= 10 numeric columns, and with m <= 10 their pair tables from one
//        pairs-only launch of the same kernel;
//      - what is left (any cardinality, m > 10 pairs, unaligned columns, the < 256-row tail):
//        keys -> 16-bit codes once, then count / sum passes over column subsets and pair passes
//        over runs of pair tables (LDS), one launch per pair table too big for LDS (u32 cells in
//        HBM); columns past the 16-bit code cache and sparse pair tables as described there.
// missed (optional): OPTIMISTIC call — the batch has not been through a dictionary pass.  The code
// translation then runs first and reports a key missing from its dictionary through *missed with
// NOTHING accumulated; the caller runs the dictionary pass and calls again without `missed`.
// Routes whose kernels read raw keys (the single-launch kernel, batches of several pieces) set
// *missed right away.
cofactor_status cat_accumulate(cofactor_agg *a, const NumCols &num, const CatCols &cat, uint64_t rows,
                               bool timed = true, const uint8_t *mask = nullptr, bool *missed = nullptr) {
  if (missed) *missed = false;
  cofactor_ctx *ctx = a->ctx;
  hipStream_t st = ctx->stream;
  const CatLayout &L = a->L;
  std::vector<CatPass> passes;
  CatPass hbm{};
  bool hbm_needed = false;
  plan_cat_passes(L, ctx->lds_budget, passes, hbm, hbm_needed);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  auto make_events = [&]() -> cofactor_status {     // (only on a path that records both)
    if (ctx->profiling && timed) {
      HIP_TRY(hipEventCreate(&e0));
      HIP_TRY(hipEventCreate(&e1));
      ctx->cat_ev.emplace_back(e0, e1);
    }
    return COFACTOR_OK;
  };
  // pair tables kept as sorted lists: sort + merge per piece of the batch (sparse.hip)
  const bool with_sparse = L.kind == 0 && any_sparse_pair(L);
  auto sparse_step = [&](const CatCols &pc, const uint8_t *pmask, uint64_t prows) -> cofactor_status {
    if (!with_sparse) return COFACTOR_OK;
    a->sparse.resize(tri(L.m));
    int q = 0;
    for (int c1 = 0; c1 < L.m; c1++)
      for (int c2 = c1; c2 < L.m; c2++, q++)
        if (pair_is_sparse(L, q)) {
          hipError_t e = sparse_add_rows(ctx->sparse_sc, a->sparse[q], pc.p[c1], pc.p[c2], pmask, prows, st);
          if (e == hipErrorInvalidValue)
            return fail(COFACTOR_ERR_UNSUPPORTED, "a sparse pair table would pass 2^31 entries");
          if (e != hipSuccess) return hip_fail(e, "sparse pair table");
        }
    return COFACTOR_OK;
  };
  // a column whose own count + sum tables exceed LDS: counts and sums with global atomics (old path)
  const bool do_s = L.kind == 0 && L.n > 0;
  bool sums_fit = true;
  for (int c = 0; c < L.m; c++) sums_fit = sums_fit && cat_sums_lds_bytes(L, 1u << c, do_s) <= ctx->lds_budget;
  // what the one-hot MFMA kernel can take: triple kind, every column <= 16 keys
  const int groups_n = (L.n + 9) / 10, groups_m = (L.m + 9) / 10;
  bool small_keys = L.kind == 0 && ctx->allow_fused && rows >= 16 * FUSED_TILE_ROWS && !ctx->no_sub;
  for (int c = 0; c < L.m; c++) small_keys = small_keys && a->nkeys_host[c] <= 16 && L.kc[c] == 16;
  // (more than 10 key columns: their pair tables need the code cache anyway, and the per-key sums then come
  //  from it too — cat_sums_mfma_kernel over <= 12 key columns x all numeric columns reads less than the
  //  sub-launches over 10 x 10 groups of raw keys)
  const bool mfma_sums = small_keys && do_s && sums_fit && L.m <= 10 &&
                         fused2_sub_fits((L.n + groups_n - 1) / groups_n, (L.m + groups_m - 1) / groups_m, mask != nullptr, L,
                                         ctx->lds_max);
  CatLayout Lp = L;                                // the key columns alone: the pairs-only launch
  Lp.n = 0; Lp.n_s = 0;
  const bool mfma_pairs = small_keys && (mfma_sums || !do_s) && L.m <= 10 &&
                          fused2_applicable(Lp, a->nkeys_host, mask != nullptr, ctx->lds_max);
  // more than 10 key columns whose per-key sums the matrix-core kernel takes: the code-cache route even
  // when every table would fit ONE launch of the LDS-atomic kernel (n x m fp64 LDS atomics per row)
  bool wide_mfma = do_s && sums_fit && L.m > 10 && ctx->allow_fused && !ctx->no_sub;
  for (int c = 0; c < L.m; c++) wide_mfma = wide_mfma && cat_sums_mfma_applicable(L, 1u << c, rows);
  const uint64_t piece = 1ull << 27;               // rows per code-cache fill / per sort
  if (missed && (rows > piece || with_sparse || (passes.size() == 1 && !hbm_needed && !mfma_sums && !mfma_pairs && !wide_mfma))) {
    *missed = true;                                  // (not a route the optimistic translation covers)
    return COFACTOR_OK;
  }
  if (passes.size() == 1 && !hbm_needed && !mfma_sums && !mfma_pairs && !wide_mfma) {   // one launch does all dense tables
    {
      cofactor_status es = make_events();
      if (es != COFACTOR_OK) return es;
    }
    if (launch_cat_accumulate(num, cat, rows, L, a->D, passes[0], true, ctx->cat_grid, st, e0, e1, mask) != hipSuccess)
      return hip_fail(hipGetLastError(), "cat_accumulate");
    for (uint64_t off = 0; with_sparse && off < rows; off += piece) {
      CatCols pc = cat;
      for (int c = 0; c < L.m; c++) pc.p[c] = cat.p[c] + off;
      cofactor_status s = sparse_step(pc, mask ? mask + off : nullptr, std::min(piece, rows - off));
      if (s != COFACTOR_OK) return s;
    }
    return COFACTOR_OK;
  }
  if (missed) {
    // (the optimistic checks below can still give up before anything is timed: the event pair is
    // made by the non-optimistic call that follows a miss, or here once the batch is known to go on)
    bool al = (reinterpret_cast<uintptr_t>(mask) & 3) == 0;
    for (int k = 0; k < L.n; k++) al = al && (reinterpret_cast<uintptr_t>(num.p[k]) & 15) == 0;
    for (int c = 0; c < L.m; c++) al = al && (reinterpret_cast<uintptr_t>(cat.p[c]) & 15) == 0;
    const uint64_t body0 = ((mfma_sums || mfma_pairs) && al) ? rows - rows % FUSED_TILE_ROWS : 0;
    const uint64_t s0 = (mfma_sums || (mfma_pairs && !do_s)) ? body0 : 0, p0 = mfma_pairs ? body0 : 0;
    if (std::min(s0, p0) != 0) { *missed = true; return COFACTOR_OK; }    // (the one-hot kernels would meet raw keys)
  }
  {
    cofactor_status es = make_events();
    if (es != COFACTOR_OK) return es;
  }
  if (e0) HIP_TRY(hipEventRecord(e0, st));
  for (uint64_t off = 0; off < rows; off += piece) {
    // (a column of the code cache: whole 64-row tiles, CODE_NONE behind the last row — catsums.hip reads tiles)
    const uint64_t prows = std::min(piece, rows - off), stride = (prows + 63) / 64 * 64;
    NumCols pn = num;
    CatCols pc = cat;
    for (int k = 0; k < L.n; k++) pn.p[k] = num.p[k] + off;
    for (int c = 0; c < L.m; c++) pc.p[c] = cat.p[c] + off;
    const uint8_t *pmask = mask ? mask + off : nullptr;
    // ---- whole tiles of aligned columns through the one-hot MFMA kernel ----
    uint64_t body = 0;
    if (mfma_sums || mfma_pairs) {
      bool al = (reinterpret_cast<uintptr_t>(pmask) & 3) == 0;
      for (int k = 0; k < L.n; k++) al = al && (reinterpret_cast<uintptr_t>(pn.p[k]) & 15) == 0;
      for (int c = 0; c < L.m; c++) al = al && (reinterpret_cast<uintptr_t>(pc.p[c]) & 15) == 0;
      if (al) body = prows - prows % FUSED_TILE_ROWS;
    }
    // first row the code-cache kernels handle: counts + sums / pairs (without per-key sums the pairs
    // launch has already counted the keys of the body)
    const uint64_t s_from = (mfma_sums || (mfma_pairs && !do_s)) ? body : 0;
    const uint64_t p_from = mfma_pairs ? body : 0;
    // ---- keys -> 16-bit codes for the rows the code-cache kernels handle ----
    const uint64_t c_from = std::min(s_from, p_from);
    cofactor_status s = COFACTOR_OK;
    if (missed && c_from != 0) {                     // (ruled out above; kept as a guard)
      if (e1) HIP_TRY(hipEventRecord(e1, st));
      *missed = true;
      return COFACTOR_OK;
    }
    if (c_from < prows) {
      s = scratch_reserve(ctx, ctx->code_cache, ctx->code_cache_bytes, (size_t)L.m * stride * 2);
      if (s != COFACTOR_OK) return s;
      CatCols cc = pc;
      for (int c = 0; c < L.m; c++) cc.p[c] = pc.p[c] + c_from;
      if (missed) HIP_TRY(hipMemsetAsync(a->D.flags + 3, 0, sizeof(int32_t), st));
      HIP_TRY(launch_cat_codes(cc, prows - c_from, stride, L, a->D, pmask ? pmask + c_from : nullptr,
                               ctx->code_cache + c_from, st, missed != nullptr));
      if (missed) {
        int32_t miss = 0;
        HIP_TRY(hipMemcpyAsync(&miss, a->D.flags + 3, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (miss) {
          if (e1) HIP_TRY(hipEventRecord(e1, st));   // (the pair stays valid: it timed the translation)
          *missed = true;
          return COFACTOR_OK;
        }
      }
    }
    if (body) {
      const int grid = fused2_grid(ctx->cus, ctx->gram_grid, body);
      const int sub_grid = !mfma_sums ? grid
          : fused2_grid(ctx->cus, ctx->gram_grid, body,
                        fused2_sub_wgs_per_cu((L.n + groups_n - 1) / groups_n, (L.m + groups_m - 1) / groups_m,
                                              pmask != nullptr, L, ctx->lds_max));
      if (mfma_pairs) {                            // pair tables, key counts, diagonal cells
        cofactor_status s = ensure_pair_slabs(ctx, fused_slab_bytes(Lp, grid));
        if (s != COFACTOR_OK) return s;
        HIP_TRY(launch_fused2(NumCols{}, pc, body, Lp, a->D, grid, ctx->lds_max, ctx->partials, ctx->pair_slabs, nullptr,
                              a->d_acc, st, nullptr, nullptr, pmask, nullptr));
      }
      if (mfma_sums)
        for (int gm = 0, c0 = 0; gm < groups_m; gm++) {
          const int mg = L.m / groups_m + (gm < L.m % groups_m ? 1 : 0);
          int idx[10];
          for (int c = 0; c < mg; c++) idx[c] = c0 + c;
          for (int gn = 0, k0 = 0; gn < groups_n; gn++) {
            const int ng = L.n / groups_n + (gn < L.n % groups_n ? 1 : 0);
            HIP_TRY(launch_fused2_sub(pn, pc, body, L, a->D, k0, ng, idx, mg, /*do_cnt=*/gn == 0 && !mfma_pairs, sub_grid,
                                      ctx->lds_max, pmask, st));
            k0 += ng;
          }
          c0 += mg;
        }
    }
    // ---- the rest on the codes ----
    if (s_from < prows) {
      NumCols tn = pn;
      CatCols tc = pc;
      for (int k = 0; k < L.n; k++) tn.p[k] = pn.p[k] + s_from;
      for (int c = 0; c < L.m; c++) tc.p[c] = pc.p[c] + s_from;
      const unsigned short *tcodes = ctx->code_cache + s_from;
      const uint64_t trows = prows - s_from;
      if (sums_fit) {
        unsigned subs[COFACTOR_MAX_CAT];
        int nsub = 0;
        unsigned sub = 0;
        size_t used = 0;
        // columns the matrix-core kernel takes (catsums.hip) go twelve to a launch
        bool mfma_cols = do_s && stride % 64 == 0;
        for (int c = 0; c < L.m; c++) mfma_cols = mfma_cols && cat_sums_mfma_applicable(L, 1u << c, trows);
        bool all_small = true;                       // <= 16 codes everywhere: up to 24 columns a launch (eight waves)
        for (int c = 0; c < L.m; c++) all_small = all_small && L.kc[c] <= 16;
        int in_sub = 0;
        for (int c = 0; c < L.m; c++) {
          const size_t b = cat_sums_lds_bytes(L, 1u << c, do_s);
          if (sub && (used + b > ctx->lds_budget || (mfma_cols && in_sub == (all_small ? 24 : 12)))) { subs[nsub++] = sub; sub = 0; used = 0; in_sub = 0; }
          sub |= 1u << c; used += b; in_sub++;
        }
        if (sub) subs[nsub++] = sub;
        if (nsub == 1 || mfma_cols)
          for (int i = 0; i < nsub; i++) HIP_TRY(launch_cat_sums(tn, tcodes, trows, stride, L, a->D, subs[i], ctx->cat_grid, st));
        else HIP_TRY(launch_cat_sums_subsets(tn, tcodes, trows, stride, L, a->D, subs, nsub, ctx->cus, st));
      } else {
        CatPass base{};
        base.do_cnt = 1; base.do_s = L.kind == 0;
        base.col_mask = (L.m >= 32) ? 0xFFFFFFFFu : ((1u << L.m) - 1u);
        base.dict_lds = hbm.dict_lds;
        HIP_TRY(launch_cat_accumulate(tn, tc, trows, L, a->D, base, false, ctx->cat_grid, st, nullptr, nullptr,
                                      pmask ? pmask + s_from : nullptr));
      }
    }
    if (L.kind == 0 && p_from < prows) {
      // pair tables: runs that fit LDS together, then the big ones one by one
      const unsigned short *tcodes = ctx->code_cache + p_from;
      const uint64_t trows = prows - p_from;
      NumCols tn = pn;
      CatCols tc = pc;
      for (int k = 0; k < L.n; k++) tn.p[k] = pn.p[k] + p_from;
      for (int c = 0; c < L.m; c++) tc.p[c] = pc.p[c] + p_from;
      const uint8_t *tmask = pmask ? pmask + p_from : nullptr;
      const int npairs = L.m * (L.m + 1) / 2;
      CatPass run{};
      size_t used = 0;
      std::vector<int> big;
      auto flush = [&]() -> hipError_t {
        if (run.p_cells == 0) return hipSuccess;
        hipError_t e = launch_cat_pairs(tcodes, trows, stride, L, a->D, run, nullptr, ctx->cat_grid, st);
        run = CatPass{};
        used = 0;
        return e;
      };
      // (a column with more than 32768 codes does not fit the 16-bit code cache: its dense pair
      // tables are updated from the raw keys by the one-launch kernel, global atomics)
      CatPass wide{};
      bool any_wide = false;
      wide.dict_lds = hbm.dict_lds;
      int q = 0;
      for (int c1 = 0; c1 < L.m; c1++)
        for (int c2 = c1; c2 < L.m && q < npairs; c2++, q++) {
          if (pair_is_sparse(L, q)) continue;
          if (L.kc[c1] > 32768 || L.kc[c2] > 32768) {
            HIP_TRY(flush());                     // (its cells interrupt the run of LDS tables)
            wide.pair_mask[q >> 5] |= 1u << (q & 31);
            wide.col_mask |= (1u << c1) | (1u << c2);
            any_wide = true;
            continue;
          }
          const size_t bytes = (size_t)L.kc[c1] * (size_t)L.kc[c2] * 4;
          if (bytes > ctx->lds_budget) { HIP_TRY(flush()); big.push_back(q); continue; }
          if (used + bytes > ctx->lds_budget) HIP_TRY(flush());
          if (run.p_cells == 0) run.p_base = L.p_off[q];
          run.pair_mask[q >> 5] |= 1u << (q & 31);
          run.p_cells += (int)(bytes / 4);
          used += bytes;
        }
      HIP_TRY(flush());
      // the big ones: rows binned by the high bits of code 1, slices of the table in LDS (cat.hip);
      // what that does not take (more than 1024 bins) goes one table at a time with global atomics
      if (ctx->allow_binned && !big.empty()) {
        BinPlan plan{};
        std::vector<int> left;
        size_t max_part = 0;
        for (int c1 = 0; c1 < L.m; c1++) {
          std::vector<int> mine;                    // this column's big pairs
          for (int qb : big) {
            int a1 = 0, rem = qb;
            while (rem >= L.m - a1) { rem -= L.m - a1; a1++; }
            if (a1 == c1) mine.push_back(qb);
          }
          if (mine.empty()) continue;
          int kc2max = 0;
          for (int qb : mine) kc2max = std::max(kc2max, L.kc[c1 + (qb - (c1 * L.m - c1 * (c1 - 1) / 2))]);
          int shift = 0;
          while ((size_t)(2 << shift) * kc2max * 4 <= ctx->lds_budget && (2 << shift) <= L.kc[c1]) shift++;
          const int nb = L.kc[c1] >> shift;
          if ((size_t)(1 << shift) * kc2max * 4 > ctx->lds_budget || nb > 1024 || nb < 1) { left.insert(left.end(), mine.begin(), mine.end()); continue; }
          const int j = plan.ncols++;
          plan.col[j] = c1; plan.shift[j] = shift; plan.nb[j] = nb; plan.npart[j] = 0;
          for (int qb : mine) {
            const int c2 = c1 + (qb - (c1 * L.m - c1 * (c1 - 1) / 2));
            const int k = plan.npart[j]++;
            plan.part[j][k] = c2; plan.part_kc[j][k] = L.kc[c2]; plan.part_poff[j][k] = L.p_off[qb];
          }
          max_part = std::max(max_part, (size_t)plan.npart[j]);
        }
        if (plan.ncols > 0) {
          if (!ctx->bin_words) {
            HIP_TRY(hipMalloc((void **)&ctx->bin_words, bin_scratch_words() * 4));
            HIP_TRY(hipMemsetAsync(ctx->bin_words, 0, bin_scratch_words() * 4, st));
            HIP_TRY(hipMalloc((void **)&ctx->bin_plan, sizeof(BinPlan)));
          }
          s = scratch_reserve(ctx, ctx->bin_codes, ctx->bin_codes_bytes, (1 + max_part) * stride * 2);
          if (s != COFACTOR_OK) return s;
          HIP_TRY(hipMemcpyAsync(ctx->bin_plan, &plan, sizeof(BinPlan), hipMemcpyHostToDevice, st));
          HIP_TRY(hipStreamSynchronize(st));         // (`plan` lives on this stack frame)
          HIP_TRY(launch_cat_binned_pairs(tcodes, trows, stride, plan, ctx->bin_plan, ctx->bin_words, ctx->bin_codes, stride,
                                          ctx->cus, a->D.p, st));
        }
        big.swap(left);
      }
      for (int qb : big) {
        int c1 = 0, rem = qb;
        while (rem >= L.m - c1) { rem -= L.m - c1; c1++; }
        const int c2 = c1 + rem;
        const size_t cells = (size_t)L.kc[c1] * (size_t)L.kc[c2];
        s = scratch_reserve(ctx, ctx->pair_tmp, ctx->pair_tmp_bytes, cells * 4);
        if (s != COFACTOR_OK) return s;
        HIP_TRY(hipMemsetAsync(ctx->pair_tmp, 0, cells * 4, st));
        CatPass one{};
        one.pair_mask[qb >> 5] |= 1u << (qb & 31);
        HIP_TRY(launch_cat_pairs(tcodes, trows, stride, L, a->D, one, ctx->pair_tmp, ctx->cat_grid, st));
        HIP_TRY(launch_cat_fold_u32(ctx->pair_tmp, (long long)cells, a->D.p + L.p_off[qb], st));
      }
      if (any_wide)
        HIP_TRY(launch_cat_accumulate(tn, tc, trows, L, a->D, wide, false, ctx->cat_grid, st, nullptr, nullptr, tmask));
    }
    s = sparse_step(pc, pmask, prows);
    if (s != COFACTOR_OK) return s;
  }
  if (e1) HIP_TRY(hipEventRecord(e1, st));
  return COFACTOR_OK;
}

cofactor_status update_device_impl(cofactor_agg *a, const NumCols &num, const CatCols &cat,
                                   uint64_t rows, bool allow_optimistic = true,
                                   const uint8_t *mask = nullptr) {
  if (rows == 0) return COFACTOR_OK;
  a->blob_cache_valid = false;
  a->dev_dirty = true;
  cofactor_ctx *ctx = a->ctx;
  hipStream_t st = ctx->stream;
  // one fused2 launch takes a bounded number of rows (its int32 pair accumulators): longer updates
  // go in pieces of whole tiles
  {
    const uint64_t piece = fused2_max_rows(ctx->cus) / FUSED_TILE_ROWS * FUSED_TILE_ROWS;
    if (a->m > 0 && rows > piece) {
      for (uint64_t off = 0; off < rows; off += piece) {
        NumCols pn = num;
        CatCols pc = cat;
        for (int k = 0; k < a->n; k++) pn.p[k] = num.p[k] + off;
        for (int c = 0; c < a->m; c++) pc.p[c] = cat.p[c] + off;
        cofactor_status s = update_device_impl(a, pn, pc, std::min(piece, rows - off), allow_optimistic,
                                               mask ? mask + off : nullptr);
        if (s != COFACTOR_OK) return s;
      }
      return COFACTOR_OK;
    }
  }
  bool aligned = rows >= FUSED_TILE_ROWS &&         // the fused kernel wants whole tiles of aligned columns
                 (reinterpret_cast<uintptr_t>(mask) & 3) == 0;
  for (int k = 0; k < a->n; k++) aligned = aligned && (reinterpret_cast<uintptr_t>(num.p[k]) & 15) == 0;
  for (int c = 0; c < a->m; c++) aligned = aligned && (reinterpret_cast<uintptr_t>(cat.p[c]) & 15) == 0;
  const uint64_t main_rows = rows - rows % FUSED_TILE_ROWS;
  // which one-pass kernel: fused_kernel (three teams, LDS-atomic pair counts) where it applies,
  // fused2_kernel (LDS-DMA ring, everything on MFMA) for what it does not take: NB aggregates, n = 0
  // fused3_kernel (the same ring with specialised pair / sum waves, two per SIMD) for the triple kind
  // with n >= 1 and m >= 2
  // nb_ring_kernel for the NB kind (no matrix product at all: sums, squares, counter increments)
  bool v1 = false, v3 = false, vnb = false;
  auto fused_fits = [&]() {
    const int pref = ctx->fused_pref;
    vnb = a->kind == COFACTOR_NB && (pref == 0 || pref == 4) && nbring_applicable(a->L, mask != nullptr, ctx->lds_max);
    if (vnb) { v1 = v3 = false; return true; }
    // measured (tests/tools/onepass_compare.py, 5e7 rows): fused3 wins from 8 key columns on (10_10: 1.78 vs
    // 1.83 ms, 2_10: 1.25 vs 1.58), fused_kernel below (10_4: 1.06 vs 1.18); COFACTOR_FUSED=3 pins it
    const bool ok3 = (pref == 3 || (pref == 0 && a->m >= 8)) &&
                     fused3_applicable(a->L, a->nkeys_host, mask != nullptr, ctx->lds_max);
    const bool ok1 = !ok3 && pref != 2 && fused_applicable(a->L, a->nkeys_host, ctx->lds_max, nullptr);
    const bool ok2 = pref != 1 && fused2_applicable(a->L, a->nkeys_host, mask != nullptr, ctx->lds_max);
    v3 = ok3;
    v1 = ok1;
    return ok1 || ok2 || ok3;
  };

  // Optimistic mode: when every column already has a dictionary that fits the fused kernel, skip
  // the dictionary pass; the kernel leaves out (and lists) the row blocks that meet an unknown key,
  // and only those are redone below after a dictionary pass over them.
  bool optimistic = allow_optimistic && ctx->allow_fused && ctx->allow_optimistic && a->m > 0 && aligned &&
                    a->cat_ready;
  if (optimistic) {
    for (int c = 0; c < a->m; c++) optimistic = optimistic && a->nkeys_host[c] >= 1;
    optimistic = optimistic && fused_fits();
  }
  bool fused = optimistic;
  // The same idea for the code-cache route: when every column has a dictionary and no one-pass kernel
  // takes the shape, the dictionary pass is skipped and the code translation (which reads every key
  // anyway) reports a miss before anything is accumulated.
  bool generic_opt = false;
  if (a->m > 0 && !optimistic) {
    generic_opt = allow_optimistic && ctx->allow_optimistic && a->cat_ready && !(ctx->allow_fused && aligned && fused_fits());
    for (int c = 0; c < a->m && generic_opt; c++) generic_opt = a->nkeys_host[c] >= 1;
    if (!generic_opt) {
      cofactor_status s = cat_dictionaries(a, cat, rows);
      if (s != COFACTOR_OK) return s;
      fused = ctx->allow_fused && aligned && fused_fits();
    }
  }
#ifdef COFACTOR_DEV_ABLATE
  if (a->m > 0) {
    int32_t mask = (int32_t)env_long("COFACTOR_CAT_ABLATE", 0);
    HIP_TRY(hipMemcpyAsync(a->D.flags + 2, &mask, sizeof(mask), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
#endif
  uint64_t done = 0;
  unsigned skipped = 0;
  const uint64_t skip_unit = v1 ? FUSED_TILE_ROWS : FUSED2_SKIP_UNIT;   // rows per entry of the skip list
  if (fused) {                                    // one pass: dense + categorical
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->profiling) {
      HIP_TRY(hipEventCreate(&e0));
      HIP_TRY(hipEventCreate(&e1));
      ctx->fused_ev.emplace_back(e0, e1);
    }
    const int grid = (v3 || vnb) ? fused2_grid(ctx->cus, ctx->gram_grid, main_rows, 1)
                   : v1 ? fused_grid(a->L, ctx->cus, ctx->gram_grid, main_rows)
                        : fused2_grid(ctx->cus, ctx->gram_grid, main_rows,
                                      fused2_wgs_per_cu(a->L, mask != nullptr, ctx->lds_max));
    {
      cofactor_status s = ensure_pair_slabs(ctx, fused_slab_bytes(a->L, grid));
      if (s != COFACTOR_OK) return s;
    }
    unsigned *skip = nullptr;
    if (optimistic) {
      const size_t need = (main_rows / skip_unit + 1) * sizeof(unsigned);
      if (need > ctx->skip_bytes) {
        HIP_TRY(hipStreamSynchronize(st));
        (void)hipFree(ctx->skip);
        ctx->skip = nullptr;
        ctx->skip_bytes = 0;
        HIP_TRY(hipMalloc((void **)&ctx->skip, need));
        ctx->skip_bytes = need;
      }
      skip = ctx->skip;
      HIP_TRY(hipMemsetAsync(skip, 0, sizeof(unsigned), st));
    }
    ctx->last_fused = vnb ? "nb_ring_kernel" : v3 ? "fused3_kernel" : (v1 ? "fused_kernel" : "fused2_kernel");
    if (vnb)
      HIP_TRY(launch_nbring(num, cat, main_rows, a->L, a->D, grid, ctx->lds_max, ctx->partials, skip, a->d_acc, st, e0, e1,
                            mask, a->d_kept));
    else if (v3)
      HIP_TRY(launch_fused3(num, cat, main_rows, a->L, a->D, grid, ctx->lds_max, ctx->partials, ctx->pair_slabs,
                            skip, a->d_acc, st, e0, e1, mask, a->d_kept));
    else if (v1)
      HIP_TRY(launch_fused(num, cat, main_rows, a->L, a->D, grid, ctx->partials, ctx->pair_slabs, skip,
                           a->d_acc, st, e0, e1, mask, a->d_kept));
    else
      HIP_TRY(launch_fused2(num, cat, main_rows, a->L, a->D, grid, ctx->lds_max, ctx->partials, ctx->pair_slabs,
                            skip, a->d_acc, st, e0, e1, mask, a->d_kept));
    done = main_rows;
    if (optimistic) {
      HIP_TRY(hipMemcpyAsync(&skipped, skip, sizeof(unsigned), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
    }
  }
  if (skipped > 0) {                              // redo the left-out row blocks, dictionary pass included
    const uint64_t trows = (uint64_t)skipped * skip_unit;
    unsigned *temp = nullptr;
    HIP_TRY(hipMalloc((void **)&temp, sizeof(unsigned) * trows * (size_t)(a->n + a->m) + (mask ? trows : 0)));
    uint8_t *tmask = mask ? reinterpret_cast<uint8_t *>(temp + trows * (size_t)(a->n + a->m)) : nullptr;
    hipError_t ge = v1 ? launch_gather_tiles(num, cat, a->n, a->m, ctx->skip + 1, skipped, temp, trows, st, mask, tmask)
                       : launch_gather_units(num, cat, a->n, a->m, (int)skip_unit, ctx->skip + 1, skipped, temp, trows,
                                             st, mask, tmask);
    cofactor_status s = ge == hipSuccess ? COFACTOR_OK : hip_fail(ge, "gather of the left-out rows");
    if (s == COFACTOR_OK) {
      NumCols gnum{};
      CatCols gcat{};
      for (int k = 0; k < a->n; k++) gnum.p[k] = reinterpret_cast<const float *>(temp) + (size_t)k * trows;
      for (int c = 0; c < a->m; c++) gcat.p[c] = reinterpret_cast<const int32_t *>(temp) + (size_t)(a->n + c) * trows;
      const double before = a->dev_rows;
      s = update_device_impl(a, gnum, gcat, trows, /*allow_optimistic=*/false, tmask);
      a->dev_rows = before;                       // these rows are counted once, below (a filter's kept
                                                  // rows were counted on the device by that call)
    }
    (void)hipStreamSynchronize(st);
    (void)hipFree(temp);
    if (s != COFACTOR_OK) return s;
  }
  if (done < rows) {                              // two-kernel path (all rows, or the < 256-row tail)
    NumCols tnum = num;
    CatCols tcat = cat;
    for (int k = 0; k < a->n; k++) tnum.p[k] = num.p[k] + done;
    for (int c = 0; c < a->m; c++) tcat.p[c] = cat.p[c] + done;
    const uint64_t trows = rows - done;
    if (optimistic) {                             // the tail's keys have not been through a dictionary pass
      cofactor_status s = cat_dictionaries(a, tcat, trows);
      if (s != COFACTOR_OK) return s;
    }
    if (a->n > 0) {
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (ctx->profiling && !fused) {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        ctx->gram_ev.emplace_back(e0, e1);
      }
      HIP_TRY(launch_gram(tnum, a->n, trows, ctx->gram_grid, ctx->partials, a->d_acc, st, e0, e1,
                          mask ? mask + done : nullptr));
    }
    if (a->m > 0) {
      bool missed = false;
      cofactor_status s = cat_accumulate(a, tnum, tcat, trows, /*timed=*/!fused, mask ? mask + done : nullptr,
                                         generic_opt ? &missed : nullptr);
      if (s != COFACTOR_OK) return s;
      if (missed) {                                 // a new key (or a route without the check): dictionary pass, then again
        s = cat_dictionaries(a, tcat, trows);
        if (s != COFACTOR_OK) return s;
        s = cat_accumulate(a, tnum, tcat, trows, /*timed=*/!fused, mask ? mask + done : nullptr);
        if (s != COFACTOR_OK) return s;
      }
    }
  }
  if (a->m > 0) a->cat_check_pending = true;      // flags[1] is looked at by the next snapshot
  if (mask) {                                     // the fused kernel counted its own kept rows
    if (done < rows) HIP_TRY(launch_count_mask(mask + done, rows - done, a->d_kept, st));
  } else {
    a->dev_rows += (double)rows;
  }
  return COFACTOR_OK;
}

// Hands the state's staging block to the context's pool (or frees it when the pool is full).
// The caller has made sure no copy or kernel still uses it.
static void stage_release(cofactor_agg *a) {
  if (!a->stage_cap) return;
  cofactor_ctx *ctx = a->ctx;
  const size_t bytes = (size_t)(a->n + a->m) * 2 * a->stage_cap * 4;
  if (ctx->stage_pool_bytes + bytes <= ((size_t)4 << 30) && ctx->stage_pool.size() < 256) {
    ctx->stage_pool.push_back({a->n, a->m, a->stage_cap, a->h_num, a->d_num, a->h_cat, a->d_cat});
    ctx->stage_pool_bytes += bytes;
  } else {
    if (a->h_num) (void)hipHostFree(a->h_num);
    if (a->h_cat) (void)hipHostFree(a->h_cat);
    (void)hipFree(a->d_num);
    (void)hipFree(a->d_cat);
  }
  a->h_num = nullptr; a->h_cat = nullptr; a->d_num = nullptr; a->d_cat = nullptr;
  a->stage_cap = 0;
}

// (Re)allocates the double-buffered staging area for `cap` rows per buffer.  Only called with
// both buffers empty on the host side; waits for the state's OWN copies and kernels still using the
// old one (not for the whole stream: other worker threads keep it busy).
cofactor_status stage_alloc(cofactor_agg *a, uint64_t cap) {
  for (int b = 0; b < 2; b++)
    if (a->stage_busy[b]) {
      HIP_TRY(hipEventSynchronize(a->stage_ev[b]));
      a->stage_busy[b] = false;
    }
  CTX_LOCK(a->ctx);
  stage_release(a);                                // (stage_cap = 0: a failure below leaves no capacity without buffers)
  cofactor_ctx *ctx = a->ctx;
  for (size_t i = 0; i < ctx->stage_pool.size(); i++) {
    const auto &blk = ctx->stage_pool[i];
    if (blk.n == a->n && blk.m == a->m && blk.cap == cap) {
      a->h_num = blk.h_num; a->d_num = blk.d_num; a->h_cat = blk.h_cat; a->d_cat = blk.d_cat;
      ctx->stage_pool_bytes -= (size_t)(a->n + a->m) * 2 * cap * 4;
      ctx->stage_pool.erase(ctx->stage_pool.begin() + (long)i);
      break;
    }
  }
  if (!a->h_num && !a->h_cat) {
    float *hn = nullptr, *dn = nullptr;
    int32_t *hc = nullptr, *dc = nullptr;
    hipError_t e = hipSuccess;
    if (a->n) {
      e = hipHostMalloc((void **)&hn, sizeof(float) * 2 * a->n * cap, hipHostMallocDefault);
      if (e == hipSuccess) e = hipMalloc((void **)&dn, sizeof(float) * 2 * a->n * cap);
    }
    if (a->m && e == hipSuccess) {
      e = hipHostMalloc((void **)&hc, sizeof(int32_t) * 2 * a->m * cap, hipHostMallocDefault);
      if (e == hipSuccess) e = hipMalloc((void **)&dc, sizeof(int32_t) * 2 * a->m * cap);
    }
    if (e != hipSuccess) {
      if (hn) (void)hipHostFree(hn);
      if (hc) (void)hipHostFree(hc);
      (void)hipFree(dn);
      (void)hipFree(dc);
      return hip_fail(e, "staging buffers");
    }
    a->h_num = hn; a->d_num = dn; a->h_cat = hc; a->d_cat = dc;
  }
  for (auto &e : a->stage_ev)
    if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync));   // (a waiting worker thread sleeps)
  a->stage_cap = cap;
  a->stage_buf = 0;
  return COFACTOR_OK;
}

static cofactor_status stage_enqueue(cofactor_agg *a);

cofactor_status stage_flush(cofactor_agg *a) {
  if (a->stage_rows == 0) return COFACTOR_OK;
  const int b = a->stage_buf;
  {
    cofactor_status s = stage_enqueue(a);          // (holds the context lock)
    if (s != COFACTOR_OK) return s;
  }
  // The buffer filled next must have left the host.  Waited for WITHOUT the context lock: other
  // worker threads keep enqueueing their own flushes meanwhile.
  if (a->stage_busy[b ^ 1]) {
    HIP_TRY(hipEventSynchronize(a->stage_ev[b ^ 1]));
    a->stage_busy[b ^ 1] = false;
  }
  return COFACTOR_OK;
}

static cofactor_status stage_enqueue(cofactor_agg *a) {
  CTX_LOCK(a->ctx);
  hipStream_t st = a->ctx->stream;
  const uint64_t cap = a->stage_cap, rows = a->stage_rows;
  const int b = a->stage_buf;
  NumCols num{};
  CatCols cat{};
  // a buffer is [column][cap]: when it is (nearly) full the columns go up in ONE copy (20 copies
  // of 1 MB reach about half of what the link gives a single 20 MB copy)
  const bool whole = rows * 8 >= cap * 7 && !a->ctx->stage_split;
  if (whole) {
    if (a->n) {
      const uint64_t off = (uint64_t)b * a->n * cap;
      HIP_TRY(hipMemcpyAsync(a->d_num + off, a->h_num + off, ((uint64_t)(a->n - 1) * cap + rows) * sizeof(float),
                             hipMemcpyHostToDevice, st));
    }
    if (a->m) {
      const uint64_t off = (uint64_t)b * a->m * cap;
      HIP_TRY(hipMemcpyAsync(a->d_cat + off, a->h_cat + off, ((uint64_t)(a->m - 1) * cap + rows) * sizeof(int32_t),
                             hipMemcpyHostToDevice, st));
    }
  }
  for (int k = 0; k < a->n; k++) {
    const uint64_t off = ((uint64_t)b * a->n + k) * cap;
    if (!whole) HIP_TRY(hipMemcpyAsync(a->d_num + off, a->h_num + off, rows * sizeof(float), hipMemcpyHostToDevice, st));
    num.p[k] = a->d_num + off;
  }
  for (int c = 0; c < a->m; c++) {
    const uint64_t off = ((uint64_t)b * a->m + c) * cap;
    if (!whole) HIP_TRY(hipMemcpyAsync(a->d_cat + off, a->h_cat + off, rows * sizeof(int32_t), hipMemcpyHostToDevice, st));
    cat.p[c] = a->d_cat + off;
  }
  cofactor_status s = update_device_impl(a, num, cat, rows);
  if (s != COFACTOR_OK) return s;
  HIP_TRY(hipEventRecord(a->stage_ev[b], st));    // buffer b is free again once this has passed
  a->stage_busy[b] = true;
  a->stage_buf = b ^ 1;
  a->stage_rows = 0;
  return COFACTOR_OK;
}

// What finalize's fast path needs to write the pair lists straight into the blob: the device
// pair tables as they are (code-indexed), the live codes of every column in ascending key order
// and their keys.
struct PairSource {
  CatLayout L;
  std::vector<unsigned long long> p;
  unsigned long long *p_pinned = nullptr;        // (large tables: pinned staging instead of p)
  std::vector<std::vector<int>> order;
  std::vector<std::vector<int32_t>> key_of;
  // sparse pairs (L.sparse_mask): the store's packed keys and counts, already in list order
  std::vector<std::vector<unsigned long long>> skeys, scnt;
  bool skip_cells = false;                       // the caller reads the pair tables on the device (finalize_on_device)
  const unsigned long long *cells() const { return p_pinned ? p_pinned : p.data(); }
  ~PairSource() { if (p_pinned) (void)hipHostFree(p_pinned); }
};

// Device tables + host accumulator -> one HostTriple (synchronises).
// pair_src (optional): the device pair tables are handed back raw instead of going into out.pair
// — for states whose host side holds no keys (no map node per entry: a 1000-key column pair has
// 1e6 of them).
cofactor_status snapshot(cofactor_agg *a, HostTriple &out, bool dense_only = false, PairSource *pair_src = nullptr) {
  CTX_LOCK(a->ctx);
  DeviceGuard guard(a->ctx->device);
  cofactor_status s = stage_flush(a);
  if (s != COFACTOR_OK) return s;
  hipStream_t st = a->ctx->stream;
  out.shape(a->kind, a->n, a->m);
  out.N = a->dev_rows;
  std::vector<double> acc(GRAM_ACC_LEN, 0.0);
  if (a->dev_dirty) {
    unsigned long long kept = 0;
    HIP_TRY(hipMemcpyAsync(&kept, a->d_kept, sizeof(kept), hipMemcpyDeviceToHost, st));
    if (a->n > 0)
      HIP_TRY(hipMemcpyAsync(acc.data(), a->d_acc, sizeof(double) * GRAM_ACC_LEN, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    out.N += (double)kept;
  }
  if (a->n > 0 && a->dev_dirty) {
    for (int c = 0; c < a->n; c++) out.lin[c] = acc[gram_lin_pos(c, a->n)];
    if (a->kind == COFACTOR_NB) {
      for (int j = 0; j < a->n; j++) out.quad[j] = acc[gram_quad_pos(j, j, a->n)];
    } else {
      size_t q = 0;
      for (int j = 0; j < a->n; j++)
        for (int k = j; k < a->n; k++) out.quad[q++] = acc[gram_quad_pos(j, k, a->n)];
    }
  }
  if (!dense_only && a->m > 0 && a->cat_ready && a->dev_dirty) {
    const CatLayout &L = a->L;
    std::vector<unsigned long long> slot(L.n_slots), cnt(L.n_cnt), p_local;
    std::vector<int32_t> code(L.n_slots);
    std::vector<double> sums(std::max(1, L.n_s));
    unsigned long long *p_dst = nullptr;
    const bool want_cells = L.n_p && !(pair_src && pair_src->skip_cells);
    if (want_cells) {
      const size_t bytes = sizeof(unsigned long long) * (size_t)L.n_p;
      if (pair_src && bytes >= ((size_t)32 << 20)) {           // big: straight into pinned memory
        HIP_TRY(hipHostMalloc((void **)&pair_src->p_pinned, bytes, hipHostMallocDefault));
        p_dst = pair_src->p_pinned;
      } else {
        std::vector<unsigned long long> &pv = pair_src ? pair_src->p : p_local;
        pv.resize(L.n_p);
        p_dst = pv.data();
      }
    }
    HIP_TRY(hipMemcpyAsync(slot.data(), a->D.ht_slot, sizeof(unsigned long long) * L.n_slots, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(code.data(), a->D.ht_code, sizeof(int32_t) * L.n_slots, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(cnt.data(), a->D.cnt, sizeof(unsigned long long) * L.n_cnt, hipMemcpyDeviceToHost, st));
    if (L.n_s) HIP_TRY(hipMemcpyAsync(sums.data(), a->D.s, sizeof(double) * L.n_s, hipMemcpyDeviceToHost, st));
    if (want_cells) HIP_TRY(hipMemcpyAsync(p_dst, a->D.p, sizeof(unsigned long long) * L.n_p, hipMemcpyDeviceToHost, st));
    int32_t flags[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(flags, a->D.flags, sizeof(flags), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (flags[1]) return fail(COFACTOR_ERR_INTERNAL, "a row met a key missing from its dictionary");
    a->cat_check_pending = false;
    std::vector<std::vector<int32_t>> key_of(a->m);
    std::vector<std::vector<char>> live(a->m);
    for (int c = 0; c < a->m; c++) {
      key_of[c].assign(L.kc[c], 0);
      live[c].assign(L.kc[c], 0);
      for (int sidx = 0; sidx < L.ht_cap[c]; sidx++) {
        const unsigned long long v = slot[L.ht_off[c] + sidx];
        const int cd = code[L.ht_off[c] + sidx];
        if (v == 0ull || cd < 0 || cd >= L.kc[c]) continue;
        const unsigned long long rows_of_key = cnt[L.cnt_off[c] + cd];
        if (rows_of_key == 0) continue;           // key known to the dictionary, not to this state
        const int32_t key = (int32_t)(unsigned)(v & 0xffffffffull);
        key_of[c][cd] = key;
        live[c][cd] = 1;
        auto &vals = out.col[c][key];
        vals.assign(a->kind ? 1 : (size_t)a->n + 1, 0.0);
        vals[0] = (double)rows_of_key;
        if (!a->kind)
          for (int k = 0; k < a->n; k++) vals[k + 1] = sums[L.s_off[c] + (size_t)cd * a->n + k];
      }
    }
    if (!a->kind) {
      // visit codes in ascending key order so that every map insertion is an append
      std::vector<std::vector<int>> order(a->m);
      for (int c = 0; c < a->m; c++) {
        for (int k = 0; k < L.kc[c]; k++)
          if (live[c][k]) order[c].push_back(k);
        std::sort(order[c].begin(), order[c].end(), [&](int x, int y) { return key_of[c][x] < key_of[c][y]; });
      }
      // sparse pair tables: sorted lists of packed keys, straight from their stores
      std::vector<std::vector<unsigned long long>> skeys(tri(a->m)), scnt(tri(a->m));
      if (any_sparse_pair(L)) {
        for (int q = 0; q < tri(a->m) && q < (int)a->sparse.size(); q++) {
          const SparseStore &sp = a->sparse[q];
          if (!pair_is_sparse(L, q) || sp.len == 0) continue;
          skeys[q].resize(sp.len);
          scnt[q].resize(sp.len);
          HIP_TRY(hipMemcpyAsync(skeys[q].data(), sp.keys, sp.len * 8, hipMemcpyDeviceToHost, st));
          HIP_TRY(hipMemcpyAsync(scnt[q].data(), sp.cnt, sp.len * 8, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(hipStreamSynchronize(st));
      }
      if (pair_src) {
        pair_src->L = L;
        pair_src->order = std::move(order);
        pair_src->key_of = std::move(key_of);
        pair_src->skeys = std::move(skeys);
        pair_src->scnt = std::move(scnt);
      } else {
        int q = 0;
        for (int c1 = 0; c1 < a->m; c1++)
          for (int c2 = c1; c2 < a->m; c2++, q++) {
            auto &tab = out.pair[q];
            if (pair_is_sparse(L, q)) {
              for (size_t i = 0; i < skeys[q].size(); i++) {
                int32_t k1, k2;
                sparse_unpack(skeys[q][i], k1, k2);
                tab.emplace_hint(tab.end(), std::make_pair(k1, k2), (double)scnt[q][i]);
              }
              continue;
            }
            for (int k1 : order[c1])
              for (int k2 : order[c2]) {
                const unsigned long long v = p_dst[L.p_off[q] + (size_t)k1 * L.kc[c2] + k2];
                if (!v) continue;
                tab.emplace_hint(tab.end(), std::make_pair(key_of[c1][k1], key_of[c2][k2]), (double)v);
              }
          }
      }
    }
  }
  if (dense_only) {
    out.N += a->host.N;
    for (int k = 0; k < a->n; k++) out.lin[k] += a->host.lin[k];
    for (size_t k = 0; k < out.quad.size(); k++) out.quad[k] += a->host.quad[k];
    return COFACTOR_OK;
  }
  std::string err;
  if (!out.add(a->host, err)) return fail(COFACTOR_ERR_INVALID, err);
  return COFACTOR_OK;
}

// Runs fn(task) for task = 0..tasks-1 on up to 32 threads (dynamic hand-out); on the calling
// thread alone when the job is small.
template <class F>
void parallel_tasks(size_t tasks, bool big, F &&fn) {
  unsigned nt = big ? std::min<unsigned>(32, std::max(1u, std::thread::hardware_concurrency())) : 1;
  if (nt > tasks) nt = (unsigned)std::max<size_t>(1, tasks);
  if (nt <= 1) {
    for (size_t t = 0; t < tasks; t++) fn(t);
    return;
  }
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    for (size_t t = next.fetch_add(1); t < tasks; t = next.fetch_add(1)) fn(t);
  };
  std::vector<std::thread> th;
  th.reserve(nt - 1);
  for (unsigned i = 1; i < nt; i++) th.emplace_back(worker);
  worker();
  for (auto &t : th) t.join();
}

// The quad_cat part of the blob from the raw pair tables: per column pair [count, (key1, key2,
// value)...] in ascending (key1, key2) order.  Two sweeps over blocks of table rows (count
// non-zero cells, then write), spread over threads: a 10-column table with 1000 keys per column
// has 5.5e7 non-zero cells.
void encode_pair_lists(const PairSource &src, int m, BlobVec &blob) {
  const CatLayout &L = src.L;
  const unsigned long long *p = src.cells();
  constexpr int RB = 64;                                      // table rows per task
  constexpr size_t SB = (size_t)1 << 16;                      // entries of a sparse store per task
  struct Task { int q, c1, c2; long long r0, r1; };
  std::vector<Task> tasks;
  std::vector<size_t> first_task(tri(m) + 1, 0);
  {
    int q = 0;
    for (int c1 = 0; c1 < m; c1++)
      for (int c2 = c1; c2 < m; c2++, q++) {
        first_task[q] = tasks.size();
        if (pair_is_sparse(L, q)) {               // entries [r0, r1) of the sorted store
          const size_t len = q < (int)src.skeys.size() ? src.skeys[q].size() : 0;
          for (size_t r = 0; r < len; r += SB) tasks.push_back({q, -1, -1, (long long)r, (long long)std::min(len, r + SB)});
          continue;
        }
        const int rows1 = (int)src.order[c1].size();
        for (int r = 0; r < rows1; r += RB) tasks.push_back({q, c1, c2, r, std::min(rows1, r + RB)});
      }
    first_task[q] = tasks.size();
  }
  size_t cells = 0;
  for (auto const &t : tasks) cells += t.c1 < 0 ? (size_t)(t.r1 - t.r0) : (size_t)(t.r1 - t.r0) * src.order[t.c2].size();
  const bool big = cells >= ((size_t)1 << 21);
  std::vector<size_t> found(tasks.size() + 1, 0);
  parallel_tasks(tasks.size(), big, [&](size_t ti) {
    const Task &t = tasks[ti];
    if (t.c1 < 0) { found[ti] = (size_t)(t.r1 - t.r0); return; }
    const auto &o1 = src.order[t.c1];
    const auto &o2 = src.order[t.c2];
    size_t nz = 0;
    for (long long r = t.r0; r < t.r1; r++) {
      const unsigned long long *row = p + L.p_off[t.q] + (size_t)o1[r] * L.kc[t.c2];
      for (int k2 : o2) nz += row[k2] != 0ull;
    }
    found[ti] = nz;
  });
  // blob offsets: every list is preceded by its length
  std::vector<size_t> at(tasks.size() + 1, 0);
  std::vector<size_t> head(tri(m), 0);
  size_t o = blob.size();
  for (int q = 0; q < tri(m); q++) {
    head[q] = o++;
    for (size_t ti = first_task[q]; ti < first_task[q + 1]; ti++) { at[ti] = o; o += 3 * found[ti]; }
  }
  blob.resize(o);                                             // (default-initialised: see BlobVec)
  for (int q = 0; q < tri(m); q++) {
    size_t cnt = 0;
    for (size_t ti = first_task[q]; ti < first_task[q + 1]; ti++) cnt += found[ti];
    blob[head[q]] = (double)cnt;
  }
  double *out = blob.data();
  parallel_tasks(tasks.size(), big, [&](size_t ti) {
    const Task &t = tasks[ti];
    double *w = out + at[ti];
    if (t.c1 < 0) {
      const auto &ks = src.skeys[t.q];
      const auto &cs = src.scnt[t.q];
      for (long long r = t.r0; r < t.r1; r++, w += 3) {
        int32_t k1, k2;
        sparse_unpack(ks[(size_t)r], k1, k2);
        w[0] = (double)k1; w[1] = (double)k2; w[2] = (double)cs[(size_t)r];
      }
      return;
    }
    const auto &o1 = src.order[t.c1];
    const auto &o2 = src.order[t.c2];
    const auto &k1s = src.key_of[t.c1];
    const auto &k2s = src.key_of[t.c2];
    for (long long r = t.r0; r < t.r1; r++) {
      const int cd1 = o1[r];
      const unsigned long long *row = p + L.p_off[t.q] + (size_t)cd1 * L.kc[t.c2];
      const double key1 = (double)k1s[cd1];
      for (int k2 : o2) {
        const unsigned long long v = row[k2];
        if (!v) continue;
        w[0] = key1; w[1] = (double)k2s[k2]; w[2] = (double)v;
        w += 3;
      }
    }
  });
}

// memcpy of a blob, on several threads when it is hundreds of MB
void copy_blob(double *dst, const double *src, size_t count) {
  constexpr size_t CH = (size_t)1 << 21;                      // 16 MB per task
  const size_t tasks = (count + CH - 1) / CH;
  parallel_tasks(tasks, count >= ((size_t)1 << 24), [&](size_t t) {
    const size_t lo = t * CH, hi = std::min(count, lo + CH);
    std::memcpy(dst + lo, src + lo, (hi - lo) * sizeof(double));
  });
}

cofactor_status emit_blob(const double *blob, size_t size, double *out, uint64_t cap, uint64_t *needed) {
  if (needed) *needed = size;
  if (!out) return COFACTOR_OK;
  if (cap < size) return fail(COFACTOR_ERR_CAPACITY, "output buffer too small");
  copy_blob(out, blob, size);
  return COFACTOR_OK;
}

cofactor_status emit_blob(const std::vector<double> &blob, double *out, uint64_t cap, uint64_t *needed) {
  return emit_blob(blob.data(), blob.size(), out, cap, needed);
}

}  // namespace detail
}  // namespace cofactor

extern "C" {

const char *cofactor_last_error(void) { return g_err.c_str(); }
int cofactor_abi_version(void) { return COFACTOR_ABI_VERSION; }
int cofactor_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

cofactor_status cofactor_ctx_create(int device, cofactor_ctx **out) {
  if (!out) return fail(COFACTOR_ERR_INVALID, "out is null");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(COFACTOR_ERR_NO_DEVICE, std::string("no HIP device available (") +
                                            (e != hipSuccess ? hipGetErrorString(e) : "0 devices") +
                                            "); libcofactor_hip has no CPU fallback");
  if (device < 0 || device >= count) return fail(COFACTOR_ERR_NO_DEVICE, "device index out of range");
  DeviceGuard guard(device);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  auto ctx = std::make_unique<cofactor_ctx>();
  ctx->device = device;
  ctx->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  ctx->gram_grid = (int)env_long("COFACTOR_GRAM_WGS_PER_CU", 4) * ctx->cus;
  ctx->cat_grid = (int)env_long("COFACTOR_CAT_WGS_PER_CU", 2) * ctx->cus;
  ctx->lds_budget = (size_t)env_long("COFACTOR_CAT_LDS_BYTES", 150 * 1024);
  if (prop.sharedMemPerBlock > 0) ctx->lds_max = prop.sharedMemPerBlock;
  if (ctx->lds_budget > ctx->lds_max) ctx->lds_budget = ctx->lds_max;
  ctx->allow_fused = env_long("COFACTOR_NO_FUSED", 0) == 0;
  ctx->allow_optimistic = env_long("COFACTOR_NO_OPTIMISTIC", 0) == 0;
  ctx->fused_pref = (int)env_long("COFACTOR_FUSED", 0);
  ctx->no_sub = env_long("COFACTOR_NO_SUB", 0) != 0;
  ctx->groups_seg = (int)env_long("COFACTOR_GROUPS_SEG", 0);
  ctx->allow_binned = env_long("COFACTOR_NO_BINNED", 0) == 0;
  ctx->stage_split = env_long("COFACTOR_STAGE_SPLIT", 0) != 0;
  ctx->stage_rows_max = (uint64_t)std::max(512L, env_long("COFACTOR_STAGE_ROWS", 1 << 18));
  // (gram_narrow_kernel runs 16 workgroups per CU: room for that many images)
  HIP_TRY(hipMalloc((void **)&ctx->partials, sizeof(double) * (size_t)std::max(ctx->gram_grid, 16 * ctx->cus) * GRAM_ACC_LEN));
  *out = ctx.release();
  return COFACTOR_OK;
}

void cofactor_ctx_destroy(cofactor_ctx *ctx) {
  if (!ctx) return;
  DeviceGuard guard(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  (void)cofactor_ctx_profile_read(ctx, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  (void)hipFree(ctx->partials);
  (void)hipFree(ctx->pair_slabs);
  (void)hipFree(ctx->skip);
  (void)hipFree(ctx->ring_red);
  (void)hipFree(ctx->ring_scratch);
  (void)hipFree(ctx->seg_scratch);
  (void)hipFree(ctx->predict_buf);
  (void)hipFree(ctx->code_cache);
  (void)hipFree(ctx->pair_tmp);
  (void)hipFree(ctx->fin_dev);
  if (ctx->fin_host) (void)hipHostFree(ctx->fin_host);
  (void)hipFree(ctx->bin_words);
  (void)hipFree(ctx->bin_plan);
  (void)hipFree(ctx->bin_codes);
  sparse_scratch_free(ctx->sparse_sc);
  for (auto &blk : ctx->stage_pool) {
    if (blk.h_num) (void)hipHostFree(blk.h_num);
    if (blk.h_cat) (void)hipHostFree(blk.h_cat);
    (void)hipFree(blk.d_num);
    (void)hipFree(blk.d_cat);
  }
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

cofactor_status cofactor_ctx_synchronize(cofactor_ctx *ctx) {
  if (!ctx) return fail(COFACTOR_ERR_INVALID, "ctx is null");
  CTX_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return COFACTOR_OK;
}

void *cofactor_ctx_stream(cofactor_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

cofactor_status cofactor_ctx_profile_enable(cofactor_ctx *ctx, int on) {
  if (!ctx) return fail(COFACTOR_ERR_INVALID, "ctx is null");
  ctx->profiling = on != 0;
  return COFACTOR_OK;
}

const char *cofactor_ctx_profile_kernel(cofactor_ctx *ctx) { return ctx ? ctx->last_fused : ""; }

cofactor_status cofactor_ctx_profile_read(cofactor_ctx *ctx, double *gram_ms, uint64_t *gram_launches,
                                          double *cat_ms, uint64_t *cat_launches, double *fused_ms,
                                          uint64_t *fused_launches) {
  if (!ctx) return fail(COFACTOR_ERR_INVALID, "ctx is null");
  CTX_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  auto drain = [](std::vector<std::pair<hipEvent_t, hipEvent_t>> &evs, double *ms, uint64_t *cnt) {
    double total = 0;
    for (auto &p : evs) {
      float t = 0;
      if (hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) total += t;
      else (void)hipGetLastError();                 // (an unrecorded pair must not leave a sticky error behind)
      (void)hipEventDestroy(p.first);
      (void)hipEventDestroy(p.second);
    }
    if (ms) *ms = total;
    if (cnt) *cnt = evs.size();
    evs.clear();
  };
  drain(ctx->gram_ev, gram_ms, gram_launches);
  drain(ctx->cat_ev, cat_ms, cat_launches);
  drain(ctx->fused_ev, fused_ms, fused_launches);
  return COFACTOR_OK;
}

cofactor_status cofactor_ctx_calibrate(cofactor_ctx *ctx, uint64_t bytes, int reps, double *copy_gbs,
                                       double *read_gbs) {
  if (!ctx || bytes < (1u << 20) || reps < 1) return fail(COFACTOR_ERR_INVALID, "calibrate: bad arguments");
  CTX_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  void *src = nullptr, *dst = nullptr;
  HIP_TRY(hipMalloc(&src, bytes));
  hipError_t e = hipMalloc(&dst, bytes);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (e == hipSuccess) e = hipMemsetAsync(src, 0x3c, bytes, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(dst, 0, bytes, ctx->stream);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  double out[2] = {0, 0};
  const int cal_grid = 16 * ctx->cus;
  for (int mode = 0; mode < 2 && e == hipSuccess; mode++) {
    const bool copy = mode == 0;
    e = launch_calibration(src, dst, bytes, cal_grid, copy, ctx->stream);     // warm-up
    if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
    for (int r = 0; r < reps && e == hipSuccess; r++) e = launch_calibration(src, dst, bytes, cal_grid, copy, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess && ms > 0)
      out[mode] = (copy ? 2.0 : 1.0) * (double)calibration_bytes(bytes, cal_grid) * reps / (ms * 1e-3) / 1e9;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(src);
  (void)hipFree(dst);
  if (e != hipSuccess) return hip_fail(e, "calibrate");
  if (copy_gbs) *copy_gbs = out[0];
  if (read_gbs) *read_gbs = out[1];
  return COFACTOR_OK;
}

cofactor_status cofactor_agg_create(cofactor_ctx *ctx, int n_num, int n_cat, cofactor_kind kind,
                                    cofactor_agg **out) {
  if (!ctx || !out) return fail(COFACTOR_ERR_INVALID, "ctx/out is null");
  *out = nullptr;
  if (n_num < 0 || n_num > COFACTOR_MAX_NUM || n_cat < 0 || n_cat > COFACTOR_MAX_CAT)
    return fail(COFACTOR_ERR_INVALID, "column counts must be in 0..20");
  if (kind != COFACTOR_TRIPLE && kind != COFACTOR_NB) return fail(COFACTOR_ERR_INVALID, "unknown kind");
  CTX_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  auto a = std::make_unique<cofactor_agg>();
  a->ctx = ctx; a->n = n_num; a->m = n_cat; a->kind = (int)kind;
  a->host.shape(a->kind, a->n, a->m);
  HIP_TRY(hipMalloc((void **)&a->d_acc, sizeof(double) * GRAM_ACC_LEN));
  HIP_TRY(hipMemsetAsync(a->d_acc, 0, sizeof(double) * GRAM_ACC_LEN, ctx->stream));
  HIP_TRY(hipMalloc((void **)&a->d_kept, sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(a->d_kept, 0, sizeof(unsigned long long), ctx->stream));
  *out = a.release();
  return COFACTOR_OK;
}

void cofactor_agg_destroy(cofactor_agg *a) {
  if (!a) return;
  {
    CTX_LOCK(a->ctx);
    if (a->ctx->fin_owner == a) a->ctx->fin_owner = nullptr;   // (a later state may reuse the address)
  }
  CTX_LOCK(a->ctx);
  DeviceGuard guard(a->ctx->device);
  (void)hipStreamSynchronize(a->ctx->stream);
  (void)hipFree(a->d_acc);
  (void)hipFree(a->d_kept);
  if (a->cat_ready) cat_free(a->D);
  stage_release(a);                                // (the stream is idle: nothing uses the block)
  for (auto &e : a->stage_ev)
    if (e) (void)hipEventDestroy(e);
  (void)hipFree(a->d_host_dense);
  for (auto &sp : a->sparse) sparse_store_free(sp);
  delete a;
}

cofactor_status cofactor_agg_reset(cofactor_agg *a) {
  if (!a) return fail(COFACTOR_ERR_INVALID, "agg is null");
  CTX_LOCK(a->ctx);
  DeviceGuard guard(a->ctx->device);
  hipStream_t st = a->ctx->stream;
  a->host.clear();
  a->blob_cache_valid = false;
  a->dev_rows = 0;
  a->dev_dirty = false;
  a->stage_rows = 0;
  HIP_TRY(hipMemsetAsync(a->d_acc, 0, sizeof(double) * GRAM_ACC_LEN, st));
  HIP_TRY(hipMemsetAsync(a->d_kept, 0, sizeof(unsigned long long), st));
  if (a->cat_ready) {
    HIP_TRY(hipMemsetAsync(a->D.cnt, 0, sizeof(unsigned long long) * std::max(1, a->L.n_cnt), st));
    HIP_TRY(hipMemsetAsync(a->D.s, 0, sizeof(double) * std::max(1, a->L.n_s), st));
    HIP_TRY(hipMemsetAsync(a->D.p, 0, sizeof(unsigned long long) * std::max(1, a->L.n_p), st));
  }
  for (auto &sp : a->sparse) sp.len = 0;
  return COFACTOR_OK;
}

cofactor_status cofactor_agg_update_device(cofactor_agg *a, const float *const *d_num,
                                           const int32_t *const *d_cat, uint64_t rows) {
  if (!a) return fail(COFACTOR_ERR_INVALID, "agg is null");
  CTX_LOCK(a->ctx);
  if ((a->n > 0 && !d_num) || (a->m > 0 && !d_cat)) return fail(COFACTOR_ERR_INVALID, "column array is null");
  NumCols num{};
  CatCols cat{};
  for (int k = 0; k < a->n; k++) {
    if (!d_num[k] && rows) return fail(COFACTOR_ERR_INVALID, "numeric column pointer is null");
    if (reinterpret_cast<uintptr_t>(d_num[k]) & 3) return fail(COFACTOR_ERR_INVALID, "numeric column is not 4-byte aligned");
    num.p[k] = d_num[k];
  }
  for (int c = 0; c < a->m; c++) {
    if (!d_cat[c] && rows) return fail(COFACTOR_ERR_INVALID, "categorical column pointer is null");
    if (reinterpret_cast<uintptr_t>(d_cat[c]) & 3) return fail(COFACTOR_ERR_INVALID, "categorical column is not 4-byte aligned");
    cat.p[c] = d_cat[c];
  }
  DeviceGuard guard(a->ctx->device);
  cofactor_status s = stage_flush(a);             // keep row order: staged host rows first
  if (s != COFACTOR_OK) return s;
  return update_device_impl(a, num, cat, rows);
}

cofactor_status cofactor_agg_update_device_masked(cofactor_agg *a, const float *const *d_num,
                                                  const int32_t *const *d_cat, const uint8_t *d_mask,
                                                  uint64_t rows) {
  if (!a) return fail(COFACTOR_ERR_INVALID, "agg is null");
  if (!d_mask) return cofactor_agg_update_device(a, d_num, d_cat, rows);
  CTX_LOCK(a->ctx);
  if ((a->n > 0 && !d_num) || (a->m > 0 && !d_cat)) return fail(COFACTOR_ERR_INVALID, "column array is null");
  NumCols num{};
  CatCols cat{};
  for (int k = 0; k < a->n; k++) {
    if (!d_num[k] && rows) return fail(COFACTOR_ERR_INVALID, "numeric column pointer is null");
    if (reinterpret_cast<uintptr_t>(d_num[k]) & 3) return fail(COFACTOR_ERR_INVALID, "numeric column is not 4-byte aligned");
    num.p[k] = d_num[k];
  }
  for (int c = 0; c < a->m; c++) {
    if (!d_cat[c] && rows) return fail(COFACTOR_ERR_INVALID, "categorical column pointer is null");
    if (reinterpret_cast<uintptr_t>(d_cat[c]) & 3) return fail(COFACTOR_ERR_INVALID, "categorical column is not 4-byte aligned");
    cat.p[c] = d_cat[c];
  }
  DeviceGuard guard(a->ctx->device);
  cofactor_status s = stage_flush(a);
  if (s != COFACTOR_OK) return s;
  return update_device_impl(a, num, cat, rows, /*allow_optimistic=*/true, d_mask);
}

cofactor_status cofactor_agg_update_host(cofactor_agg *a, const float *const *num,
                                         const int32_t *const *cat, const uint32_t *const *num_sel,
                                         const uint32_t *const *cat_sel, const uint32_t *row_idx,
                                         uint64_t rows) {
  if (!a) return fail(COFACTOR_ERR_INVALID, "agg is null");
  if ((a->n > 0 && !num) || (a->m > 0 && !cat)) return fail(COFACTOR_ERR_INVALID, "column array is null");
  if (rows == 0) return COFACTOR_OK;
  a->blob_cache_valid = false;
  DeviceGuard guard(a->ctx->device);
  // Staging starts small and grows with the rows a state actually receives (x8 per full buffer up
  // to COFACTOR_STAGE_ROWS): a GROUP BY with thousands of states must not pin 2 x 20 MB for each.
  if (!a->stage_cap) {
    const uint64_t max_rows = a->ctx->stage_rows_max;
    cofactor_status s = stage_alloc(a, std::min<uint64_t>(max_rows, 512));
    if (s != COFACTOR_OK) return s;
  }
  uint64_t done = 0;
  while (done < rows) {
    const uint64_t take = std::min(rows - done, a->stage_cap - a->stage_rows);
    for (int k = 0; k < a->n; k++) {
      float *dst = a->h_num + ((uint64_t)a->stage_buf * a->n + k) * a->stage_cap + a->stage_rows;
      const float *src = num[k];
      const uint32_t *sel = num_sel ? num_sel[k] : nullptr;
      if (!sel && !row_idx) std::memcpy(dst, src + done, take * sizeof(float));
      else
        for (uint64_t i = 0; i < take; i++) {
          const uint64_t r = row_idx ? row_idx[done + i] : done + i;
          dst[i] = src[sel ? sel[r] : r];
        }
    }
    for (int c = 0; c < a->m; c++) {
      int32_t *dst = a->h_cat + ((uint64_t)a->stage_buf * a->m + c) * a->stage_cap + a->stage_rows;
      const int32_t *src = cat[c];
      const uint32_t *sel = cat_sel ? cat_sel[c] : nullptr;
      if (!sel && !row_idx) std::memcpy(dst, src + done, take * sizeof(int32_t));
      else
        for (uint64_t i = 0; i < take; i++) {
          const uint64_t r = row_idx ? row_idx[done + i] : done + i;
          dst[i] = src[sel ? sel[r] : r];
        }
    }
    a->stage_rows += take;
    done += take;
    if (a->stage_rows == a->stage_cap) {
      cofactor_status s = stage_flush(a);
      if (s != COFACTOR_OK) return s;
      const uint64_t max_rows = a->ctx->stage_rows_max;
      if (a->stage_cap < max_rows) {
        s = stage_alloc(a, std::min(max_rows, a->stage_cap * 8));
        if (s != COFACTOR_OK) return s;
      }
    }
  }
  return COFACTOR_OK;
}

cofactor_status cofactor_agg_update_triples(cofactor_agg *a, const double *blobs,
                                            const uint64_t *offsets, uint64_t count) {
  if (!a || (count && (!blobs || !offsets))) return fail(COFACTOR_ERR_INVALID, "null argument");
  a->blob_cache_valid = false;
  ListTriple t;
  std::string err;
  for (uint64_t i = 0; i < count; i++) {
    if (offsets[i + 1] < offsets[i]) return fail(COFACTOR_ERR_INVALID, "update_triples: offsets must ascend");
    if (!blob_decode(blobs + offsets[i], offsets[i + 1] - offsets[i], t, err)) return fail(COFACTOR_ERR_INVALID, err);
    if (!a->host.add_list(t, err)) return fail(COFACTOR_ERR_INVALID, err);
  }
  return COFACTOR_OK;
}

// finalize of a state with millions of dense pair cells: everything before quad_cat is encoded on the
// host from the small tables as usual; the quad_cat lists are written by kernels (cat.hip) straight
// into their place in a device image of the blob and land in a pinned host buffer of the context.
static cofactor_status finalize_on_device(cofactor_agg *a) {
  cofactor_ctx *ctx = a->ctx;
  DeviceGuard guard(ctx->device);
  hipStream_t st = ctx->stream;
  HostTriple snap;
  PairSource pairs;
  pairs.skip_cells = true;
  cofactor_status s = snapshot(a, snap, false, &pairs);
  if (s != COFACTOR_OK) return s;
  if ((int)pairs.order.size() != a->m) return fail(COFACTOR_ERR_INTERNAL, "finalize: pair source incomplete");
  std::vector<double> head;
  snap.encode_without_pairs(head);
  const CatLayout &L = pairs.L;
  const int m = a->m, np = tri(m);
  // flat arrays: live codes of every column in ascending key order, code -> key per column
  std::vector<int> order_flat, order_off(m + 1, 0), key_flat(std::max(1, L.n_cnt), 0);
  for (int c = 0; c < m; c++) {
    order_flat.insert(order_flat.end(), pairs.order[c].begin(), pairs.order[c].end());
    order_off[c + 1] = (int)order_flat.size();
    for (int k = 0; k < L.kc[c]; k++) key_flat[L.cnt_off[c] + k] = pairs.key_of[c][k];
  }
  std::vector<PairListInfo> info(np);
  std::vector<int> row_pair, row_code;
  std::vector<size_t> first_row(np + 1, 0);
  {
    int q = 0;
    for (int c1 = 0; c1 < m; c1++)
      for (int c2 = c1; c2 < m; c2++, q++) {
        info[q] = PairListInfo{(long long)L.p_off[q], L.kc[c2], order_off[c2], order_off[c2 + 1] - order_off[c2], L.cnt_off[c1], L.cnt_off[c2]};
        first_row[q] = row_pair.size();
        for (int code : pairs.order[c1]) { row_pair.push_back(q); row_code.push_back(code); }
      }
    first_row[np] = row_pair.size();
  }
  const int rows = (int)row_pair.size();
  // device copies of the small arrays (one allocation)
  const size_t ints = order_flat.size() + key_flat.size() + 2 * (size_t)rows + 16;
  const size_t bytes = ints * 4 + info.size() * sizeof(PairListInfo) + (size_t)rows * (4 + 8) + 64;
  unsigned char *d_small = nullptr;
  HIP_TRY(hipMalloc((void **)&d_small, bytes));
  auto fail_free = [&](cofactor_status st_) { (void)hipStreamSynchronize(st); (void)hipFree(d_small); return st_; };
  unsigned char *cur = d_small;
  auto place = [&](const void *src, size_t n) -> void * {
    void *at = cur;
    if (n && src && hipMemcpyAsync(at, src, n, hipMemcpyHostToDevice, st) != hipSuccess) return nullptr;
    cur += (n + 15) / 16 * 16;
    return at;
  };
  PairListInfo *d_info = (PairListInfo *)place(info.data(), info.size() * sizeof(PairListInfo));
  int *d_order = (int *)place(order_flat.data(), order_flat.size() * 4);
  int *d_key = (int *)place(key_flat.data(), key_flat.size() * 4);
  int *d_rp = (int *)place(row_pair.data(), (size_t)rows * 4);
  int *d_rc = (int *)place(row_code.data(), (size_t)rows * 4);
  unsigned *d_cnt = (unsigned *)place(nullptr, (size_t)rows * 4);
  unsigned long long *d_base = (unsigned long long *)place(nullptr, (size_t)rows * 8);
  if (!d_info || !d_order || !d_key || !d_rp || !d_rc || cur > d_small + bytes + 256) return fail_free(hip_fail(hipGetLastError(), "finalize: upload"));
  if (launch_pairlist_count(a->D.p, d_rp, d_rc, rows, d_info, d_order, d_cnt, st) != hipSuccess) return fail_free(hip_fail(hipGetLastError(), "pairlist_count"));
  std::vector<unsigned> rowcnt(std::max(1, rows));
  if (rows && hipMemcpyAsync(rowcnt.data(), d_cnt, (size_t)rows * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return fail_free(hip_fail(hipGetLastError(), "finalize"));
  if (hipStreamSynchronize(st) != hipSuccess) return fail_free(hip_fail(hipGetLastError(), "finalize"));
  // layout of the pair section: per pair [length, triples...]
  std::vector<unsigned long long> rowbase(std::max(1, rows));
  std::vector<size_t> list_at(np), list_len(np, 0);
  size_t o = 0;
  for (int q = 0; q < np; q++) {
    list_at[q] = o++;
    for (size_t t = first_row[q]; t < first_row[q + 1]; t++) { rowbase[t] = o; o += 3 * (size_t)rowcnt[t]; list_len[q] += rowcnt[t]; }
  }
  const size_t total = head.size() + o;
  if (o > ctx->fin_dev_cap) {
    (void)hipFree(ctx->fin_dev); ctx->fin_dev = nullptr; ctx->fin_dev_cap = 0;
    if (hipMalloc((void **)&ctx->fin_dev, o * 8) != hipSuccess) return fail_free(hip_fail(hipGetLastError(), "finalize: device blob"));
    ctx->fin_dev_cap = o;
  }
  if (total > ctx->fin_host_cap) {
    if (ctx->fin_host) (void)hipHostFree(ctx->fin_host);
    ctx->fin_host = nullptr; ctx->fin_host_cap = 0; ctx->fin_owner = nullptr;
    const size_t want = total + total / 8;
    if (hipHostMalloc((void **)&ctx->fin_host, want * 8, hipHostMallocDefault) != hipSuccess) return fail_free(hip_fail(hipGetLastError(), "finalize: pinned blob"));
    ctx->fin_host_cap = want;
  }
  ctx->fin_owner = nullptr;
  if (rows && hipMemcpyAsync(d_base, rowbase.data(), (size_t)rows * 8, hipMemcpyHostToDevice, st) != hipSuccess) return fail_free(hip_fail(hipGetLastError(), "finalize"));
  if (launch_pairlist_fill(a->D.p, d_rp, d_rc, rows, d_info, d_order, d_key, d_base, ctx->fin_dev, st) != hipSuccess) return fail_free(hip_fail(hipGetLastError(), "pairlist_fill"));
  if (o && hipMemcpyAsync(ctx->fin_host + head.size(), ctx->fin_dev, o * 8, hipMemcpyDeviceToHost, st) != hipSuccess) return fail_free(hip_fail(hipGetLastError(), "finalize: D2H"));
  std::memcpy(ctx->fin_host, head.data(), head.size() * 8);         // (meanwhile: the part before quad_cat)
  if (hipStreamSynchronize(st) != hipSuccess) return fail_free(hip_fail(hipGetLastError(), "finalize"));
  for (int q = 0; q < np; q++) ctx->fin_host[head.size() + list_at[q]] = (double)list_len[q];
  (void)hipFree(d_small);
  ctx->fin_len = total;
  ctx->fin_owner = a;
  a->blob_in_ctx = true;
  a->blob_cache_valid = true;
  return COFACTOR_OK;
}

cofactor_status cofactor_agg_finalize(cofactor_agg *a, double *out, uint64_t cap, uint64_t *needed) {
  if (!a) return fail(COFACTOR_ERR_INVALID, "agg is null");
  CTX_LOCK(a->ctx);
  if (a->blob_cache_valid && a->blob_in_ctx && a->ctx->fin_owner != a) a->blob_cache_valid = false;   // (another state's blob took the buffer)
  if (!a->blob_cache_valid || a->stage_rows > 0) {
    a->blob_in_ctx = false;
    {
      bool host_keys0 = false;
      for (auto const &c : a->host.col) host_keys0 = host_keys0 || !c.empty();
      for (auto const &t : a->host.pair) host_keys0 = host_keys0 || !t.empty();
      // millions of dense pair cells: the lists are written on the device
      if (a->kind == COFACTOR_TRIPLE && a->m > 0 && !host_keys0 && a->cat_ready && a->dev_dirty && !any_sparse_pair(a->L) &&
          a->L.n_p >= (1 << 22) && a->ctx->allow_binned) {
        cofactor_status fs = stage_flush(a);
        if (fs != COFACTOR_OK) return fs;
        fs = finalize_on_device(a);
        if (fs != COFACTOR_OK) return fs;
        return emit_blob(a->ctx->fin_host, a->ctx->fin_len, out, cap, needed);
      }
    }
    HostTriple snap;
    bool host_has_keys = false;       // keys merged in on the host (combine, lifted triples)
    for (auto const &c : a->host.col) host_has_keys = host_has_keys || !c.empty();
    for (auto const &t : a->host.pair) host_has_keys = host_has_keys || !t.empty();
    PairSource pairs;
    const bool direct = a->kind == COFACTOR_TRIPLE && a->m > 0 && !host_has_keys;
    cofactor_status s = snapshot(a, snap, false, direct ? &pairs : nullptr);
    if (s != COFACTOR_OK) return s;
    std::vector<double> head;
    if (direct && (int)pairs.order.size() == a->m) {
      snap.encode_without_pairs(head);
      a->blob_cache.assign(head.begin(), head.end());
      encode_pair_lists(pairs, a->m, a->blob_cache);
    } else {
      snap.encode(head);
      a->blob_cache.assign(head.begin(), head.end());
    }
    a->blob_cache_valid = true;
  }
  if (a->blob_in_ctx) return emit_blob(a->ctx->fin_host, a->ctx->fin_len, out, cap, needed);
  return emit_blob(a->blob_cache.data(), a->blob_cache.size(), out, cap, needed);
}

uint64_t cofactor_dense_len(int n_num, cofactor_kind kind) {
  return 1 + (uint64_t)n_num + (kind == COFACTOR_NB ? (uint64_t)n_num : tri(n_num));
}

cofactor_status cofactor_agg_export_dense_device(cofactor_agg *a, double *d_out) {
  if (!a || !d_out) return fail(COFACTOR_ERR_INVALID, "null argument");
  CTX_LOCK(a->ctx);
  DeviceGuard guard(a->ctx->device);
  cofactor_status s = stage_flush(a);             // rows still staged on the host belong to the totals
  if (s != COFACTOR_OK) return s;
  hipStream_t st = a->ctx->stream;
  // whatever the state holds on the host (combine, lifted triples) rides along as an addend
  const uint64_t len = cofactor_dense_len(a->n, (cofactor_kind)a->kind);
  bool host_dense = a->host.N != 0;
  for (double v : a->host.lin) host_dense = host_dense || v != 0;
  for (double v : a->host.quad) host_dense = host_dense || v != 0;
  const double *extra = nullptr;
  if (host_dense) {
    if (!a->d_host_dense) HIP_TRY(hipMalloc((void **)&a->d_host_dense, sizeof(double) * 256));
    a->host_dense_stage.assign(1, a->host.N);
    a->host_dense_stage.insert(a->host_dense_stage.end(), a->host.lin.begin(), a->host.lin.end());
    a->host_dense_stage.insert(a->host_dense_stage.end(), a->host.quad.begin(), a->host.quad.end());
    HIP_TRY(hipMemcpyAsync(a->d_host_dense, a->host_dense_stage.data(), len * sizeof(double),
                           hipMemcpyHostToDevice, st));
    extra = a->d_host_dense;
  }
  HIP_TRY(launch_dense_export(a->d_acc, a->d_kept, a->dev_rows, extra, a->n, a->kind, d_out, st));
  return COFACTOR_OK;
}

cofactor_status cofactor_agg_import_dense_device(cofactor_agg *a, const double *d_in) {
  if (!a || !d_in) return fail(COFACTOR_ERR_INVALID, "null argument");
  CTX_LOCK(a->ctx);
  DeviceGuard guard(a->ctx->device);
  cofactor_status s = stage_flush(a);
  if (s != COFACTOR_OK) return s;
  // the accumulator image and the kept-row counter become the imported totals; the host-side
  // dense addends are part of those totals now.  Categorical tables are not touched.
  HIP_TRY(launch_dense_import(d_in, a->n, a->kind, a->d_acc, a->d_kept, a->ctx->stream));
  a->dev_rows = 0;
  a->dev_dirty = true;
  a->blob_cache_valid = false;
  a->host.N = 0;
  std::fill(a->host.lin.begin(), a->host.lin.end(), 0.0);
  std::fill(a->host.quad.begin(), a->host.quad.end(), 0.0);
  return COFACTOR_OK;
}

// ---- dictionary-aligned table seam ----------------------------------------

namespace {

// every key a state knows, per column, ascending: device dictionary + host-side maps
cofactor_status collect_keys(cofactor_agg *a, std::vector<std::vector<int32_t>> &keys,
                             std::vector<unsigned long long> *slots_out = nullptr,
                             std::vector<int32_t> *codes_out = nullptr) {
  keys.assign(a->m, {});
  if (a->m > 0 && a->cat_ready) {
    hipStream_t st = a->ctx->stream;
    const CatLayout &L = a->L;
    std::vector<unsigned long long> slot(L.n_slots);
    std::vector<int32_t> code(L.n_slots);
    HIP_TRY(hipMemcpyAsync(slot.data(), a->D.ht_slot, sizeof(unsigned long long) * L.n_slots, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(code.data(), a->D.ht_code, sizeof(int32_t) * L.n_slots, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int c = 0; c < a->m; c++)
      for (int i = 0; i < L.ht_cap[c]; i++) {
        const unsigned long long v = slot[L.ht_off[c] + i];
        if (v != 0ull && code[L.ht_off[c] + i] >= 0) keys[c].push_back((int32_t)(unsigned)(v & 0xffffffffull));
      }
    if (slots_out) *slots_out = std::move(slot);
    if (codes_out) *codes_out = std::move(code);
  }
  for (int c = 0; c < a->m; c++) {
    for (auto const &kv : a->host.col[c]) keys[c].push_back(kv.first);
    std::sort(keys[c].begin(), keys[c].end());
    keys[c].erase(std::unique(keys[c].begin(), keys[c].end()), keys[c].end());
  }
  return COFACTOR_OK;
}

uint64_t keys_signature(const std::vector<std::vector<int32_t>> &keys) {
  uint64_t h = 0xcbf29ce484222325ull;               // FNV-1a over (column separator, keys)
  auto mix = [&](uint64_t v) { for (int b = 0; b < 8; b++) { h ^= (v >> (8 * b)) & 0xff; h *= 0x100000001b3ull; } };
  for (auto const &col : keys) {
    mix(0xffffffffffffffffull ^ col.size());
    for (int32_t k : col) mix((uint64_t)(uint32_t)k);
  }
  return h ? h : 1;
}

}  // namespace

cofactor_status cofactor_agg_keys(cofactor_agg *a, int32_t *out, uint64_t cap, uint64_t *needed,
                                  uint64_t *offsets) {
  if (!a) return fail(COFACTOR_ERR_INVALID, "agg is null");
  CTX_LOCK(a->ctx);
  DeviceGuard guard(a->ctx->device);
  cofactor_status s = stage_flush(a);
  if (s != COFACTOR_OK) return s;
  std::vector<std::vector<int32_t>> keys;
  s = collect_keys(a, keys);
  if (s != COFACTOR_OK) return s;
  uint64_t total = 0;
  for (auto const &k : keys) total += k.size();
  if (needed) *needed = total;
  if (!out) return COFACTOR_OK;
  if (cap < total) return fail(COFACTOR_ERR_CAPACITY, "output buffer too small");
  uint64_t pos = 0;
  for (int c = 0; c < a->m; c++) {
    if (offsets) offsets[c] = pos;
    std::copy(keys[c].begin(), keys[c].end(), out + pos);
    pos += keys[c].size();
  }
  if (offsets) offsets[a->m] = pos;
  return COFACTOR_OK;
}

cofactor_status cofactor_agg_dict_signature(cofactor_agg *a, uint64_t *sig) {
  if (!a || !sig) return fail(COFACTOR_ERR_INVALID, "null argument");
  CTX_LOCK(a->ctx);
  bool host_keys = false;
  for (auto const &c : a->host.col) host_keys = host_keys || !c.empty();
  for (auto const &t : a->host.pair) host_keys = host_keys || !t.empty();
  *sig = (a->stage_rows > 0 || host_keys) ? 0 : a->dict_sig;
  return COFACTOR_OK;
}

cofactor_status cofactor_agg_align_keys(cofactor_agg *a, const int32_t *keys_in, const uint64_t *offsets) {
  if (!a || (a->m > 0 && !offsets)) return fail(COFACTOR_ERR_INVALID, "null argument");
  if (a->m == 0) return COFACTOR_OK;
  CTX_LOCK(a->ctx);
  DeviceGuard guard(a->ctx->device);
  cofactor_status s = stage_flush(a);
  if (s != COFACTOR_OK) return s;
  s = cat_prepare(a);
  if (s != COFACTOR_OK) return s;
  hipStream_t st = a->ctx->stream;
  // the global key lists: whatever was handed in (any order, duplicates allowed: the
  // concatenation of all ranks' lists), sorted and made unique here
  std::vector<std::vector<int32_t>> G(a->m);
  for (int c = 0; c < a->m; c++) {
    if (offsets[c + 1] < offsets[c]) return fail(COFACTOR_ERR_INVALID, "align_keys: offsets must ascend");
    if (offsets[c + 1] > offsets[c] && !keys_in) return fail(COFACTOR_ERR_INVALID, "null argument");
    G[c].assign(keys_in + offsets[c], keys_in + offsets[c + 1]);
    std::sort(G[c].begin(), G[c].end());
    G[c].erase(std::unique(G[c].begin(), G[c].end()), G[c].end());
    if (G[c].size() > (1u << 27)) return fail(COFACTOR_ERR_UNSUPPORTED, "categorical column has too many distinct keys");
  }
  std::vector<std::vector<int32_t>> own;
  std::vector<unsigned long long> old_slot;
  std::vector<int32_t> old_code;
  s = collect_keys(a, own, &old_slot, &old_code);
  if (s != COFACTOR_OK) return s;
  for (int c = 0; c < a->m; c++)
    if (!std::includes(G[c].begin(), G[c].end(), own[c].begin(), own[c].end()))
      return fail(COFACTOR_ERR_INVALID, "align_keys: the key lists lack a key this state holds");
  const CatLayout Lo = a->L;
  CatLayout Ln{};
  Ln.n = a->n; Ln.m = a->m; Ln.kind = a->kind;
  for (int c = 0; c < a->m; c++) {
    Ln.kc[c] = std::max(16, next_pow2((int)G[c].size()));
    Ln.ht_cap[c] = std::max(64, next_pow2(2 * (int)G[c].size()));
  }
  // (pair tables too big to be dense stay / become sorted lists: they are keyed by the keys
  // themselves, so alignment does not touch them)
  if (!cat_finish_layout(Ln, /*allow_sparse=*/a->kind == COFACTOR_TRIPLE))
    return fail(COFACTOR_ERR_UNSUPPORTED, "categorical cardinalities too high for dense code-indexed pair tables");
  if (any_sparse_pair(Ln)) a->sparse.resize(tri(a->m));
  // a pair table that is dense now and sparse under the aligned layout moves into its store first
  if (a->kind == COFACTOR_TRIPLE && any_sparse_pair(Ln) && a->dev_dirty) {
    int32_t *key_of = nullptr;
    hipError_t e2 = hipSuccess;
    int q = 0;
    for (int c1 = 0; c1 < a->m && e2 == hipSuccess; c1++)
      for (int c2 = c1; c2 < a->m && e2 == hipSuccess; c2++, q++) {
        if (!pair_is_sparse(Ln, q) || pair_is_sparse(Lo, q)) continue;
        if (!key_of) {
          e2 = hipMalloc((void **)&key_of, sizeof(int32_t) * std::max(1, Lo.n_cnt));
          for (int c = 0; c < a->m && e2 == hipSuccess; c++)
            e2 = launch_key_of_code(a->D.ht_slot + Lo.ht_off[c], a->D.ht_code + Lo.ht_off[c], Lo.ht_cap[c], Lo.kc[c],
                                    key_of + Lo.cnt_off[c], st);
        }
        if (e2 == hipSuccess)
          e2 = sparse_add_dense(a->ctx->sparse_sc, a->sparse[q], a->D.p + Lo.p_off[q], Lo.kc[c1], Lo.kc[c2],
                                key_of + Lo.cnt_off[c1], key_of + Lo.cnt_off[c2], st);
      }
    if (key_of) { (void)hipStreamSynchronize(st); (void)hipFree(key_of); }
    if (e2 != hipSuccess) return hip_fail(e2, "align_keys: dense -> sparse pair table");
  }
  // aligned dictionary (code = rank of the key in the global list) and old code -> new code
  std::vector<unsigned long long> nslot(Ln.n_slots, 0ull);
  std::vector<int32_t> ncode(Ln.n_slots, -1), remap(std::max(1, Lo.n_cnt), -1);
  for (int c = 0; c < a->m; c++) {
    const int cap = Ln.ht_cap[c];
    for (size_t g = 0; g < G[c].size(); g++) {
      unsigned h = cat_hash_key(G[c][g], cap);
      while (nslot[Ln.ht_off[c] + h] != 0ull) h = (h + 1) & (cap - 1);
      nslot[Ln.ht_off[c] + h] = (1ull << 32) | (unsigned long long)(unsigned)G[c][g];
      ncode[Ln.ht_off[c] + h] = (int32_t)g;
    }
    for (int i = 0; i < Lo.ht_cap[c]; i++) {
      const unsigned long long v = old_slot[Lo.ht_off[c] + i];
      const int32_t cd = old_code[Lo.ht_off[c] + i];
      if (v == 0ull || cd < 0 || cd >= Lo.kc[c]) continue;
      const int32_t key = (int32_t)(unsigned)(v & 0xffffffffull);
      remap[Lo.cnt_off[c] + cd] = (int32_t)(std::lower_bound(G[c].begin(), G[c].end(), key) - G[c].begin());
    }
  }
  CatDevice Dn;
  s = cat_alloc(Ln, Dn, false, st);
  if (s != COFACTOR_OK) return s;
  Dn.nkeys = a->D.nkeys;
  Dn.flags = a->D.flags;
  int32_t *d_remap = nullptr;
  int32_t nk[COFACTOR_MAX_CAT] = {0};
  for (int c = 0; c < a->m; c++) nk[c] = (int32_t)G[c].size();
  hipError_t e = hipMalloc((void **)&d_remap, sizeof(int32_t) * remap.size());
  if (e == hipSuccess) e = hipMemcpyAsync(d_remap, remap.data(), sizeof(int32_t) * remap.size(), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(Dn.ht_slot, nslot.data(), sizeof(unsigned long long) * Ln.n_slots, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(Dn.ht_code, ncode.data(), sizeof(int32_t) * Ln.n_slots, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(Dn.nkeys, nk, sizeof(nk), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = launch_cat_remap(Lo, a->D, Ln, Dn, d_remap, st);
  // what the state holds on the host under keys goes into the (now dense and aligned) tables
  bool host_keys = false;
  for (auto const &c : a->host.col) host_keys = host_keys || !c.empty();
  for (auto const &t : a->host.pair) host_keys = host_keys || !t.empty();
  double *d_add = nullptr;
  std::vector<double> addv;
  if (e == hipSuccess && host_keys) {
    addv.assign((size_t)Ln.n_cnt + Ln.n_s + Ln.n_p, 0.0);
    auto code_of = [&](int c, int32_t key) { return (size_t)(std::lower_bound(G[c].begin(), G[c].end(), key) - G[c].begin()); };
    for (int c = 0; c < a->m; c++)
      for (auto const &kv : a->host.col[c]) {
        const size_t cd = code_of(c, kv.first);
        addv[Ln.cnt_off[c] + cd] += kv.second[0];
        if (!a->kind)
          for (int k = 0; k < a->n; k++) addv[(size_t)Ln.n_cnt + Ln.s_off[c] + cd * a->n + k] += kv.second[k + 1];
      }
    if (!a->kind) {
      int q = 0;
      for (int c1 = 0; c1 < a->m; c1++)
        for (int c2 = c1; c2 < a->m; c2++, q++)
          for (auto const &kv : a->host.pair[q]) {
            if (pair_is_sparse(Ln, q)) continue;    // (merged into the pair's sorted store below)
            // a pair's keys are keys of their columns' lin_cat lists (same rows), hence in G
            const size_t k1 = code_of(c1, kv.first.first), k2 = code_of(c2, kv.first.second);
            if (k1 >= G[c1].size() || k2 >= G[c2].size() || G[c1][k1] != kv.first.first || G[c2][k2] != kv.first.second) {
              e = hipErrorInvalidValue;
              continue;
            }
            addv[(size_t)Ln.n_cnt + Ln.n_s + Ln.p_off[q] + k1 * Ln.kc[c2] + k2] += kv.second;
          }
    }
    if (e == hipSuccess) e = hipMalloc((void **)&d_add, sizeof(double) * addv.size());
    if (e == hipSuccess) e = hipMemcpyAsync(d_add, addv.data(), sizeof(double) * addv.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = launch_cat_tables_import(Ln, Dn, d_add, /*add=*/true, st);
    if (e == hipSuccess && !a->kind && any_sparse_pair(Ln)) {   // host-side entries of sparse pairs: into their stores
      int q = 0;
      for (int c1 = 0; c1 < a->m && e == hipSuccess; c1++)
        for (int c2 = c1; c2 < a->m && e == hipSuccess; c2++, q++) {
          if (!pair_is_sparse(Ln, q) || a->host.pair[q].empty()) continue;
          std::vector<unsigned long long> hk, hv;
          for (auto const &kv : a->host.pair[q]) {
            hk.push_back(sparse_pack(kv.first.first, kv.first.second));
            hv.push_back((unsigned long long)(kv.second + 0.5));
          }
          unsigned long long *dk = nullptr;
          e = hipMalloc((void **)&dk, hk.size() * 16);
          if (e == hipSuccess) e = hipMemcpyAsync(dk, hk.data(), hk.size() * 8, hipMemcpyHostToDevice, st);
          if (e == hipSuccess) e = hipMemcpyAsync(dk + hk.size(), hv.data(), hk.size() * 8, hipMemcpyHostToDevice, st);
          if (e == hipSuccess) e = sparse_merge_lists(a->ctx->sparse_sc, a->sparse[q], dk, dk + hk.size(), hk.size(), st);
          (void)hipStreamSynchronize(st);
          (void)hipFree(dk);
        }
    }
  }
  hipError_t es = hipStreamSynchronize(st);
  if (e == hipSuccess) e = es;
  (void)hipFree(d_remap);
  (void)hipFree(d_add);
  if (e != hipSuccess) {
    Dn.nkeys = nullptr; Dn.flags = nullptr;
    cat_free(Dn);
    return e == hipErrorInvalidValue ? fail(COFACTOR_ERR_INVALID, "align_keys: a host-side pair key is missing from its column's keys")
                                     : hip_fail(e, "align_keys");
  }
  CatDevice old = a->D;
  old.nkeys = nullptr; old.flags = nullptr;
  cat_free(old);
  a->D = Dn;
  a->L = Ln;
  for (int c = 0; c < a->m; c++) a->nkeys_host[c] = nk[c];
  if (host_keys) {
    for (auto &c : a->host.col) c.clear();
    for (auto &t : a->host.pair) t.clear();
  }
  a->dev_dirty = true;
  a->blob_cache_valid = false;
  a->dict_sig = keys_signature(G);
  return COFACTOR_OK;
}

uint64_t cofactor_agg_tables_len(cofactor_agg *a) {
  if (!a || a->m == 0 || !a->cat_ready) return 0;
  return (uint64_t)a->L.n_cnt + (uint64_t)a->L.n_s + (uint64_t)a->L.n_p;
}

cofactor_status cofactor_agg_export_tables_device(cofactor_agg *a, double *d_out) {
  if (!a || !d_out) return fail(COFACTOR_ERR_INVALID, "null argument");
  if (a->m == 0 || !a->cat_ready) return COFACTOR_OK;
  CTX_LOCK(a->ctx);
  DeviceGuard guard(a->ctx->device);
  if (a->stage_rows > 0) return fail(COFACTOR_ERR_INVALID, "export_tables: rows are still staged on the host (align first)");
  HIP_TRY(launch_cat_tables_export(a->L, a->D, d_out, a->ctx->stream));
  return COFACTOR_OK;
}

cofactor_status cofactor_agg_import_tables_device(cofactor_agg *a, const double *d_in) {
  if (!a || !d_in) return fail(COFACTOR_ERR_INVALID, "null argument");
  if (a->m == 0 || !a->cat_ready) return COFACTOR_OK;
  CTX_LOCK(a->ctx);
  DeviceGuard guard(a->ctx->device);
  HIP_TRY(launch_cat_tables_import(a->L, a->D, d_in, /*add=*/false, a->ctx->stream));
  a->dev_dirty = true;
  a->blob_cache_valid = false;
  return COFACTOR_OK;
}

// ---- combine on the device (Triple::SumStateCombine, sum_state.cpp:10-114) ------------------------------
// dst += src without a host round trip of the tables: the accumulator image is added by a kernel;
// the two states' dictionaries are aligned to the union of their key lists (remap kernels; only the
// key lists themselves, a few KB, pass through the host, and not even those while both states still
// hold a common alignment), then the table image [cnt | s | p] of src is added into dst's by
// export -> (peer copy when the states live on different GPUs) -> import(add); sorted pair lists
// are merged (sort + reduce by key).  src keeps its value (its tables may be re-indexed).
namespace {

bool holds_host_keys(const cofactor_agg *a) {
  for (auto const &c : a->host.col) if (!c.empty()) return true;
  for (auto const &t : a->host.pair) if (!t.empty()) return true;
  return false;
}

// bytes on `ddev` <- bytes on `sdev`, ordered on `st` (a stream of ddev)
hipError_t copy_between(void *dst, int ddev, const void *src, int sdev, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  if (ddev == sdev) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
  return hipMemcpyPeerAsync(dst, ddev, src, sdev, bytes, st);
}

}  // namespace

cofactor_status cofactor_agg_combine(cofactor_agg *dst, cofactor_agg *src) {
  if (!dst || !src) return fail(COFACTOR_ERR_INVALID, "null argument");
  if (dst == src) return fail(COFACTOR_ERR_INVALID, "combine of a state with itself");
  if (dst->n != src->n || dst->m != src->m || dst->kind != src->kind)
    return fail(COFACTOR_ERR_INVALID, "combine: state shapes differ");
  cofactor_ctx *cd = dst->ctx, *cs = src->ctx;
  // both contexts, in address order (two threads combining a -> b and b -> a must not deadlock)
  std::unique_lock<std::recursive_mutex> l1(cd < cs ? cd->mu : cs->mu);
  std::unique_lock<std::recursive_mutex> l2;
  if (cd != cs) l2 = std::unique_lock<std::recursive_mutex>(cd < cs ? cs->mu : cd->mu);
  const int ddev = cd->device, sdev = cs->device;
  dst->blob_cache_valid = false;
  cofactor_status s;
  { DeviceGuard g(sdev); if ((s = stage_flush(src)) != COFACTOR_OK) return s; }
  { DeviceGuard g(ddev); if ((s = stage_flush(dst)) != COFACTOR_OK) return s; }
  // what src holds on the host (earlier combines of host states, lifted triples): dense addends
  // are added on the host, keys are folded into its device tables by the alignment below
  dst->host.N += src->host.N;
  for (size_t k = 0; k < dst->host.lin.size() && k < src->host.lin.size(); k++) dst->host.lin[k] += src->host.lin[k];
  for (size_t k = 0; k < dst->host.quad.size() && k < src->host.quad.size(); k++) dst->host.quad[k] += src->host.quad[k];
  const bool src_keys = src->m > 0 && ((src->cat_ready && src->dev_dirty) || holds_host_keys(src));
  hipEvent_t ev = nullptr;
  double *tmp_s = nullptr, *tmp_d = nullptr;
  unsigned long long *sp_tmp = nullptr;
  auto cleanup = [&]() {
    if (ev) (void)hipEventDestroy(ev);
    if (tmp_s) { DeviceGuard g(sdev); (void)hipFree(tmp_s); }
    if (tmp_d && tmp_d != tmp_s) { DeviceGuard g(ddev); (void)hipFree(tmp_d); }
    if (sp_tmp) { DeviceGuard g(ddev); (void)hipFree(sp_tmp); }
  };
#define COMBINE_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return hip_fail(e_, #expr); } } while (0)
  if (src_keys) {
    // ---- dictionaries: same alignment already, or align both to the union of their key lists ----
    uint64_t sig_s = 0, sig_d = 0;
    (void)cofactor_agg_dict_signature(src, &sig_s);
    (void)cofactor_agg_dict_signature(dst, &sig_d);
    if (!(sig_s != 0 && sig_s == sig_d)) {
      std::vector<std::vector<int32_t>> ks, kd;
      { DeviceGuard g(sdev); if ((s = collect_keys(src, ks)) != COFACTOR_OK) { cleanup(); return s; } }
      { DeviceGuard g(ddev); if ((s = collect_keys(dst, kd)) != COFACTOR_OK) { cleanup(); return s; } }
      std::vector<int32_t> all;
      std::vector<uint64_t> offs(dst->m + 1, 0);
      for (int c = 0; c < dst->m; c++) {
        all.insert(all.end(), ks[c].begin(), ks[c].end());
        all.insert(all.end(), kd[c].begin(), kd[c].end());
        offs[c + 1] = all.size();
      }
      if ((s = cofactor_agg_align_keys(dst, all.data(), offs.data())) != COFACTOR_OK) { cleanup(); return s; }
      if ((s = cofactor_agg_align_keys(src, all.data(), offs.data())) != COFACTOR_OK) { cleanup(); return s; }
    }
    // ---- tables: src image -> (src GPU) -> dst GPU -> added into dst's tables ----
    const CatLayout &L = dst->L;
    const size_t tlen = (size_t)L.n_cnt + L.n_s + L.n_p;
    if (src->L.n_cnt != L.n_cnt || src->L.n_s != L.n_s || src->L.n_p != L.n_p) {
      cleanup();
      return fail(COFACTOR_ERR_INTERNAL, "combine: aligned states disagree on the table layout");
    }
    {
      DeviceGuard g(sdev);
      COMBINE_TRY(hipMalloc((void **)&tmp_s, std::max<size_t>(1, tlen) * sizeof(double)));
      COMBINE_TRY(launch_cat_tables_export(src->L, src->D, tmp_s, cs->stream));
      COMBINE_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      COMBINE_TRY(hipEventRecord(ev, cs->stream));
    }
    {
      DeviceGuard g(ddev);
      COMBINE_TRY(hipStreamWaitEvent(cd->stream, ev, 0));
      tmp_d = tmp_s;
      if (ddev != sdev) {
        COMBINE_TRY(hipMalloc((void **)&tmp_d, std::max<size_t>(1, tlen) * sizeof(double)));
        COMBINE_TRY(copy_between(tmp_d, ddev, tmp_s, sdev, tlen * sizeof(double), cd->stream));
      }
      COMBINE_TRY(launch_cat_tables_import(dst->L, dst->D, tmp_d, /*add=*/true, cd->stream));
      // sorted pair lists: merge src's into dst's
      if (dst->kind == COFACTOR_TRIPLE && any_sparse_pair(L)) {
        dst->sparse.resize(tri(dst->m));
        for (int q = 0; q < tri(dst->m) && q < (int)src->sparse.size(); q++) {
          const SparseStore &sp = src->sparse[q];
          if (!pair_is_sparse(L, q) || sp.len == 0) continue;
          const unsigned long long *k = sp.keys, *v = sp.cnt;
          if (ddev != sdev) {
            COMBINE_TRY(hipStreamSynchronize(cd->stream));
            if (sp_tmp) { (void)hipFree(sp_tmp); sp_tmp = nullptr; }
            COMBINE_TRY(hipMalloc((void **)&sp_tmp, sp.len * 16));
            COMBINE_TRY(copy_between(sp_tmp, ddev, sp.keys, sdev, sp.len * 8, cd->stream));
            COMBINE_TRY(copy_between(sp_tmp + sp.len, ddev, sp.cnt, sdev, sp.len * 8, cd->stream));
            k = sp_tmp; v = sp_tmp + sp.len;
          }
          hipError_t e = sparse_merge_lists(cd->sparse_sc, dst->sparse[q], k, v, sp.len, cd->stream);
          if (e == hipErrorInvalidValue) { cleanup(); return fail(COFACTOR_ERR_UNSUPPORTED, "a sparse pair table would pass 2^31 entries"); }
          COMBINE_TRY(e);
        }
      }
    }
    dst->cat_check_pending = dst->cat_check_pending || src->cat_check_pending;
  }
  // ---- dense part: accumulator image and kept-row counter ----
  if (src->dev_dirty) {
    DeviceGuard g(ddev);
    if (!ev) {
      { DeviceGuard gs(sdev); COMBINE_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); COMBINE_TRY(hipEventRecord(ev, cs->stream)); }
      COMBINE_TRY(hipStreamWaitEvent(cd->stream, ev, 0));
    }
    const double *s_acc = src->d_acc;
    const unsigned long long *s_kept = src->d_kept;
    double *img = nullptr;
    if (ddev != sdev) {                              // image + counter over to dst's GPU first
      COMBINE_TRY(hipMalloc((void **)&img, sizeof(double) * (GRAM_ACC_LEN + 1)));
      COMBINE_TRY(copy_between(img, ddev, src->d_acc, sdev, sizeof(double) * GRAM_ACC_LEN, cd->stream));
      COMBINE_TRY(copy_between(img + GRAM_ACC_LEN, ddev, src->d_kept, sdev, sizeof(unsigned long long), cd->stream));
      s_acc = img;
      s_kept = reinterpret_cast<const unsigned long long *>(img + GRAM_ACC_LEN);
    }
    hipError_t e = launch_acc_add(s_acc, s_kept, dst->d_acc, dst->d_kept, cd->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(cd->stream);
    if (img) (void)hipFree(img);
    COMBINE_TRY(e);
    dst->dev_rows += src->dev_rows;
    dst->dev_dirty = true;
  }
  {   // src's buffers were read by dst's stream: nothing of src may change before that has passed
    DeviceGuard g(ddev);
    COMBINE_TRY(hipStreamSynchronize(cd->stream));
  }
#undef COMBINE_TRY
  cleanup();
  return COFACTOR_OK;
}

// ---- sorted pair lists across ranks (SURVEY.md §8e for pair tables too big to be dense) -----------------
// A pair kept as a sorted (key1, key2) -> count list is keyed by the keys themselves: the ranks
// gather each other's lists (variable length) and every rank merges them into its own.
cofactor_status cofactor_agg_sparse_lens(cofactor_agg *a, uint64_t *lens, uint64_t cap) {
  if (!a || !lens) return fail(COFACTOR_ERR_INVALID, "null argument");
  const uint64_t np = (uint64_t)tri(a->m);
  if (cap < np) return fail(COFACTOR_ERR_CAPACITY, "lens holds fewer than m(m+1)/2 entries");
  CTX_LOCK(a->ctx);
  for (uint64_t q = 0; q < np; q++)
    lens[q] = (a->kind == COFACTOR_TRIPLE && a->cat_ready && pair_is_sparse(a->L, (int)q) && q < a->sparse.size()) ? a->sparse[q].len : 0;
  return COFACTOR_OK;
}

cofactor_status cofactor_agg_sparse_is_list(cofactor_agg *a, int32_t pair, int32_t *is_list) {
  if (!a || !is_list || pair < 0 || pair >= tri(a->m)) return fail(COFACTOR_ERR_INVALID, "bad argument");
  CTX_LOCK(a->ctx);
  *is_list = (a->kind == COFACTOR_TRIPLE && a->cat_ready && pair_is_sparse(a->L, pair)) ? 1 : 0;
  return COFACTOR_OK;
}

cofactor_status cofactor_agg_sparse_export_device(cofactor_agg *a, int32_t pair, uint64_t *d_keys, uint64_t *d_counts) {
  if (!a || pair < 0 || pair >= tri(a->m)) return fail(COFACTOR_ERR_INVALID, "bad argument");
  CTX_LOCK(a->ctx);
  DeviceGuard guard(a->ctx->device);
  if ((size_t)pair >= a->sparse.size() || a->sparse[pair].len == 0) return COFACTOR_OK;
  if (!d_keys || !d_counts) return fail(COFACTOR_ERR_INVALID, "null argument");
  const SparseStore &sp = a->sparse[pair];
  HIP_TRY(hipMemcpyAsync(d_keys, sp.keys, sp.len * 8, hipMemcpyDeviceToDevice, a->ctx->stream));
  HIP_TRY(hipMemcpyAsync(d_counts, sp.cnt, sp.len * 8, hipMemcpyDeviceToDevice, a->ctx->stream));
  return COFACTOR_OK;
}

cofactor_status cofactor_agg_sparse_assign_device(cofactor_agg *a, int32_t pair, const uint64_t *d_keys,
                                                  const uint64_t *d_counts, uint64_t len) {
  if (!a || pair < 0 || pair >= tri(a->m) || (len && (!d_keys || !d_counts))) return fail(COFACTOR_ERR_INVALID, "bad argument");
  CTX_LOCK(a->ctx);
  DeviceGuard guard(a->ctx->device);
  if (!(a->kind == COFACTOR_TRIPLE && a->cat_ready && pair_is_sparse(a->L, pair)))
    return fail(COFACTOR_ERR_INVALID, "sparse_assign: this pair table is dense in the state's layout");
  a->sparse.resize(tri(a->m));
  SparseStore &sp = a->sparse[pair];
  HIP_TRY(hipStreamSynchronize(a->ctx->stream));
  sp.len = 0;                                       // the lists handed in REPLACE the store (they include this rank's)
  hipError_t e = sparse_merge_lists(a->ctx->sparse_sc, sp, (const unsigned long long *)d_keys,
                                    (const unsigned long long *)d_counts, (size_t)len, a->ctx->stream);
  if (e == hipErrorInvalidValue) return fail(COFACTOR_ERR_UNSUPPORTED, "a sparse pair table would pass 2^31 entries");
  if (e != hipSuccess) return hip_fail(e, "sparse_assign");
  a->dev_dirty = true;
  a->blob_cache_valid = false;
  return COFACTOR_OK;
}

cofactor_status cofactor_lift_host(const float *const *num, int n_num, const int32_t *const *cat,
                                   int n_cat, uint64_t rows, cofactor_kind kind, double *out,
                                   uint64_t cap, uint64_t *needed, uint64_t *offsets) {
  if (n_num < 0 || n_cat < 0 || (n_num > 0 && !num) || (n_cat > 0 && !cat))
    return fail(COFACTOR_ERR_INVALID, "bad column arguments");
  std::vector<double> blob;
  ListTriple t;
  for (uint64_t r = 0; r < rows; r++) {
    if (offsets) offsets[r] = blob.size();
    lift_row(num, n_num, cat, n_cat, r, (int)kind, t);
    blob_encode(t, blob);
  }
  if (offsets) offsets[rows] = blob.size();
  return emit_blob(blob, out, cap, needed);
}

cofactor_status cofactor_triple_multiply(const double *a, uint64_t a_len, const double *b, uint64_t b_len,
                                         double *out, uint64_t cap, uint64_t *needed) {
  if (!a || !b) return fail(COFACTOR_ERR_INVALID, "null argument");
  ListTriple A, B, R;
  std::string err;
  if (!blob_decode(a, a_len, A, err) || !blob_decode(b, b_len, B, err)) return fail(COFACTOR_ERR_INVALID, err);
  if (!multiply(A, B, R, err)) return fail(COFACTOR_ERR_INVALID, err);
  std::vector<double> blob;
  blob_encode(R, blob);
  return emit_blob(blob, out, cap, needed);
}

// a +- b on the flat blobs when both hold exactly the same (ascending) key lists — what a MICE loop
// adds and subtracts (the cofactor of the table and of some of its rows): no decode into lists, no
// std::map per list, no encode.  false: the structures differ (or are malformed): the general path decides.
static bool add_sub_flat(const double *a, const double *b, uint64_t len, bool sub, std::vector<double> &r) {
  if (len < 4 || a[0] != b[0] || a[1] != b[1] || a[2] != b[2]) return false;
  const double kind = a[0], nd = a[1], md = a[2];
  if ((kind != 0 && kind != 1) || nd < 0 || md < 0 || nd > 64 || md > 64 || nd != std::floor(nd) || md != std::floor(md)) return false;
  const uint64_t n = (uint64_t)nd, m = (uint64_t)md;
  const double sgn = sub ? -1.0 : 1.0;
  r.resize(len);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
  uint64_t p = 3;
  const uint64_t dense = 1 + n + (kind ? n : n * (n + 1) / 2);
  if (p + dense > len) return false;
  for (uint64_t i = 0; i < dense; i++, p++) r[p] = a[p] + sgn * b[p];
  auto lists = [&](uint64_t count, uint64_t width) {   // width: key words per entry
    for (uint64_t l = 0; l < count; l++) {
      if (p >= len || a[p] != b[p] || a[p] < 0 || a[p] != std::floor(a[p])) return false;
      const uint64_t ln = (uint64_t)a[p];
      r[p] = a[p]; p++;
      if (ln > (len - p) / (width + 1)) return false;
      for (uint64_t e = 0; e < ln; e++) {
        for (uint64_t w = 0; w < width; w++, p++) { if (a[p] != b[p]) return false; r[p] = a[p]; }
        if (e && !(a[p - width - (width + 1)] < a[p - width] ||
                   (width == 2 && a[p - width - (width + 1)] == a[p - width] && a[p - width + 1 - (width + 1)] < a[p - width + 1])))
          return false;                               // (not ascending: the general path sorts)
        r[p] = a[p] + sgn * b[p]; p++;
      }
    }
    return true;
  };
  if (!lists(m, 1)) return false;
  if (!kind && (!lists(n * m, 1) || !lists(m * (m + 1) / 2, 2))) return false;
  return p == len;
}

static cofactor_status add_sub_impl(const double *a, uint64_t a_len, const double *b, uint64_t b_len, bool sub,
                                    double *out, uint64_t cap, uint64_t *needed) {
  if (!a || !b) return fail(COFACTOR_ERR_INVALID, "null argument");
  if (a_len == b_len) {
    std::vector<double> flat;
    if (add_sub_flat(a, b, a_len, sub, flat)) { g_err.clear(); return emit_blob(flat, out, cap, needed); }
  }
  ListTriple A, B, R;
  std::string err, warn;
  if (!blob_decode(a, a_len, A, err) || !blob_decode(b, b_len, B, err)) return fail(COFACTOR_ERR_INVALID, err);
  if (A.kind != B.kind) return fail(COFACTOR_ERR_INVALID, "triple kinds differ");
  add_sub(A, B, sub, R, warn);
  g_err = warn;                                   // "" unless a subtract met an unknown key
  std::vector<double> blob;
  blob_encode(R, blob);
  return emit_blob(blob, out, cap, needed);
}

cofactor_status cofactor_triple_add(const double *a, uint64_t a_len, const double *b, uint64_t b_len,
                                    double *out, uint64_t cap, uint64_t *needed) {
  return add_sub_impl(a, a_len, b, b_len, false, out, cap, needed);
}
cofactor_status cofactor_triple_sub(const double *a, uint64_t a_len, const double *b, uint64_t b_len,
                                    double *out, uint64_t cap, uint64_t *needed) {
  return add_sub_impl(a, a_len, b, b_len, true, out, cap, needed);
}

uint64_t cofactor_blob_len(const double *blob, uint64_t cap) { return blob_len(blob, cap); }

cofactor_status cofactor_triple_to_text(const double *blob, uint64_t blob_len_, int32_t aggregate_names, char *out,
                                        uint64_t cap, uint64_t *needed) {
  if (!blob) return fail(COFACTOR_ERR_INVALID, "null argument");
  ListTriple t;
  std::string err;
  if (!blob_decode(blob, blob_len_, t, err)) return fail(COFACTOR_ERR_INVALID, err);
  const std::string text = triple_to_text(t, aggregate_names != 0);
  if (needed) *needed = text.size() + 1;
  if (!out) return COFACTOR_OK;
  if (cap < text.size() + 1) return fail(COFACTOR_ERR_CAPACITY, "output buffer too small");
  std::memcpy(out, text.c_str(), text.size() + 1);
  return COFACTOR_OK;
}

cofactor_status cofactor_triple_from_text(const char *text, uint64_t text_len, double *out, uint64_t cap,
                                          uint64_t *needed) {
  if (!text) return fail(COFACTOR_ERR_INVALID, "null argument");
  ListTriple t;
  std::string err;
  if (!triple_from_text(text, (size_t)text_len, t, err)) return fail(COFACTOR_ERR_INVALID, err);
  std::vector<double> blob;
  blob_encode(t, blob);
  return emit_blob(blob, out, cap, needed);
}

// ---- consumers of the triple ----------------------------------------------------------------------

static cofactor_status emit_floats(const std::vector<float> &v, float *out, uint64_t cap,
                                   uint64_t *needed) {
  if (needed) *needed = v.size();
  if (!out) return COFACTOR_OK;
  if (cap < v.size()) return fail(COFACTOR_ERR_CAPACITY, "output buffer too small");
  std::memcpy(out, v.data(), v.size() * sizeof(float));
  return COFACTOR_OK;
}

// The trainers follow the two-call protocol (size query, then the call with a buffer), and a size
// query has to train to learn the length.  The result of the last training of this thread is kept,
// keyed by a hash of the triple and the arguments: the second call of a pair returns it.
namespace {
struct TrainMemo {
  uint64_t key = 0;
  uint64_t len = 0;
  int which = -1;
  std::vector<float> params;
};
thread_local TrainMemo g_train_memo;

uint64_t train_key(const double *triple, uint64_t len, const void *args, size_t arg_bytes) {
  uint64_t h = 0xcbf29ce484222325ull;
  auto mix = [&](const unsigned char *p, size_t n) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; std::memcpy(&w, p + i, 8); h = (h ^ w) * 0x100000001b3ull; h ^= h >> 29; }
    for (; i < n; i++) h = (h ^ p[i]) * 0x100000001b3ull;
  };
  mix(reinterpret_cast<const unsigned char *>(triple), (size_t)len * sizeof(double));
  mix(reinterpret_cast<const unsigned char *>(args), arg_bytes);
  return h;
}
}  // namespace

cofactor_status cofactor_linreg_train(const double *triple, uint64_t triple_len, int32_t label, float step_size,
                                      float lambda, int32_t max_iterations,
                                      int32_t compute_variance, int32_t normalize, float *out,
                                      uint64_t cap, uint64_t *needed) {
  if (!triple) return fail(COFACTOR_ERR_INVALID, "null argument");
  struct { int32_t label; float step, lambda; int32_t it, var, norm; } args = {label, step_size, lambda, max_iterations,
                                                                              compute_variance != 0, normalize != 0};
  const uint64_t key = train_key(triple, triple_len, &args, sizeof(args));
  TrainMemo &memo = g_train_memo;
  if (memo.which == 0 && memo.key == key && memo.len == triple_len) return emit_floats(memo.params, out, cap, needed);
  ListTriple t;
  std::string err;
  if (!blob_decode(triple, triple_len, t, err)) return fail(COFACTOR_ERR_INVALID, err);
  std::vector<float> params;
  if (!linreg_train(t, label, step_size, lambda, max_iterations, compute_variance != 0,
                    normalize != 0, params, err))
    return fail(COFACTOR_ERR_INVALID, err);
  memo.which = 0; memo.key = key; memo.len = triple_len; memo.params = params;
  return emit_floats(params, out, cap, needed);
}

cofactor_status cofactor_lda_train(const double *triple, uint64_t triple_len, int32_t label, float shrinkage,
                                   int32_t normalize, float *out, uint64_t cap, uint64_t *needed) {
  if (!triple) return fail(COFACTOR_ERR_INVALID, "null argument");
  struct { int32_t label; float shrinkage; int32_t norm; } args = {label, shrinkage, normalize != 0};
  const uint64_t key = train_key(triple, triple_len, &args, sizeof(args));
  TrainMemo &memo = g_train_memo;
  if (memo.which == 1 && memo.key == key && memo.len == triple_len) return emit_floats(memo.params, out, cap, needed);
  ListTriple t;
  std::string err;
  if (!blob_decode(triple, triple_len, t, err)) return fail(COFACTOR_ERR_INVALID, err);
  std::vector<float> params;
  if (!lda_train(t, label, shrinkage, normalize != 0, params, err))
    return fail(COFACTOR_ERR_INVALID, err);
  memo.which = 1; memo.key = key; memo.len = triple_len; memo.params = params;
  return emit_floats(params, out, cap, needed);
}

// uploads the model, runs the kernel on the context stream and waits for it
static cofactor_status predict_device(cofactor_ctx *ctx, const PredictModel &mdl, bool argmax,
                                      bool emit_label, bool noise, uint64_t seed,
                                      const float *const *d_num, const int32_t *const *d_cat,
                                      const uint8_t *d_mask, uint64_t rows, float *out_f,
                                      int32_t *out_i, const uint32_t *d_row_ids = nullptr) {
  if (mdl.F > COFACTOR_MAX_NUM || mdl.M > COFACTOR_MAX_CAT)
    return fail(COFACTOR_ERR_UNSUPPORTED, "too many columns");
  if ((mdl.F && !d_num) || (mdl.M && !d_cat) || (rows && !out_f && !out_i))
    return fail(COFACTOR_ERR_INVALID, "null argument");
  NumCols nc{}; CatCols cc{};
  for (int i = 0; i < mdl.F; i++) {
    if (!d_num[i] && rows) return fail(COFACTOR_ERR_INVALID, "null numeric column");
    nc.p[i] = d_num[i];
  }
  for (int i = 0; i < mdl.M; i++) {
    if (!d_cat[i] && rows) return fail(COFACTOR_ERR_INVALID, "null key column");
    cc.p[i] = d_cat[i];
  }
  if (rows == 0) return COFACTOR_OK;
  CTX_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  const size_t lds_limit = 64 * 1024;
  int w_in_lds = 0;
  if (predict_lds_bytes(mdl.F, mdl.M, mdl.C, mdl.KT, lds_limit, &w_in_lds) > lds_limit)
    return fail(COFACTOR_ERR_UNSUPPORTED, "predict: the key dictionaries of the model exceed the LDS budget");
  const size_t nk = mdl.kbegin.size() + mdl.keys.size() + mdl.labels.size();
  std::vector<int32_t> hi(mdl.kbegin);
  hi.insert(hi.end(), mdl.keys.begin(), mdl.keys.end());
  hi.insert(hi.end(), mdl.labels.begin(), mdl.labels.end());
  // the model's device copy lives in a context buffer (a MICE loop predicts several row ranges per
  // column: an allocation and a free per call cost more than the kernel on 1e7 rows)
  const size_t w_off = (sizeof(int32_t) * nk + 255) & ~(size_t)255;
  {
    cofactor_status rs = scratch_reserve(ctx, ctx->predict_buf, ctx->predict_bytes, w_off + sizeof(double) * mdl.W.size() + 256);
    if (rs != COFACTOR_OK) return rs;
  }
  int32_t *d_i = reinterpret_cast<int32_t *>(ctx->predict_buf);
  double *d_w = reinterpret_cast<double *>(ctx->predict_buf + w_off);
  hipError_t e = hipMemcpyAsync(d_i, hi.data(), sizeof(int32_t) * nk, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_w, mdl.W.data(), sizeof(double) * mdl.W.size(), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    const int32_t *kb = d_i, *keys = d_i + mdl.kbegin.size(), *labels = keys + mdl.keys.size();
    e = launch_predict(argmax, nc, cc, mdl.F, mdl.M, mdl.C, mdl.KT, kb, keys, d_w, d_mask, rows,
                       out_f, out_i, (argmax && emit_label) ? labels : nullptr, noise ? 1 : 0,
                       mdl.noise_sd, seed, ctx->cus * 8, lds_limit, ctx->stream, d_row_ids);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // (`hi` and the model are the caller's / this frame's)
  if (e != hipSuccess) return fail(COFACTOR_ERR_HIP, hipGetErrorString(e));
  return COFACTOR_OK;
}

cofactor_status cofactor_linreg_predict_device(cofactor_ctx *ctx, const float *params,
                                               uint64_t n_params, int32_t noise,
                                               int32_t normalize, uint64_t seed,
                                               const float *const *d_num, int32_t n_num,
                                               const int32_t *const *d_cat, int32_t n_cat,
                                               const uint8_t *d_mask, uint64_t rows, float *d_out) {
  if (!ctx || !params || n_num < 0 || n_cat < 0) return fail(COFACTOR_ERR_INVALID, "null argument");
  PredictModel mdl;
  std::string err;
  if (!linreg_model(params, n_params, n_num, n_cat, noise != 0, normalize != 0, mdl, err))
    return fail(COFACTOR_ERR_INVALID, err);
  return predict_device(ctx, mdl, false, false, noise != 0, seed, d_num, d_cat, d_mask, rows, d_out, nullptr);
}

cofactor_status cofactor_linreg_predict_rows_device(cofactor_ctx *ctx, const float *params, uint64_t n_params,
                                                    int32_t noise, int32_t normalize, uint64_t seed,
                                                    const float *const *d_num, int32_t n_num,
                                                    const int32_t *const *d_cat, int32_t n_cat,
                                                    const uint8_t *d_mask, const uint32_t *d_row_ids, uint64_t rows,
                                                    float *d_out) {
  if (!ctx || !params || n_num < 0 || n_cat < 0) return fail(COFACTOR_ERR_INVALID, "null argument");
  PredictModel mdl;
  std::string err;
  if (!linreg_model(params, n_params, n_num, n_cat, noise != 0, normalize != 0, mdl, err))
    return fail(COFACTOR_ERR_INVALID, err);
  return predict_device(ctx, mdl, false, false, noise != 0, seed, d_num, d_cat, d_mask, rows, d_out, nullptr, d_row_ids);
}

cofactor_status cofactor_lda_predict_device(cofactor_ctx *ctx, const float *params,
                                            uint64_t n_params, int32_t normalize,
                                            int32_t emit_label, const float *const *d_num,
                                            int32_t n_num, const int32_t *const *d_cat,
                                            int32_t n_cat, const uint8_t *d_mask, uint64_t rows,
                                            int32_t *d_out) {
  if (!ctx || !params || n_num < 0 || n_cat < 0) return fail(COFACTOR_ERR_INVALID, "null argument");
  PredictModel mdl;
  std::string err;
  if (!lda_model(params, n_params, n_num, n_cat, normalize != 0, mdl, err))
    return fail(COFACTOR_ERR_INVALID, err);
  return predict_device(ctx, mdl, true, emit_label != 0, false, 0, d_num, d_cat, d_mask, rows, nullptr, d_out);
}

// host columns: one packed device buffer [F floats columns | M key columns | out], staged per call
static cofactor_status predict_host(cofactor_ctx *ctx, const PredictModel &mdl, bool argmax,
                                    bool emit_label, bool noise, uint64_t seed,
                                    const float *const *num, const int32_t *const *cat,
                                    uint64_t rows, void *out) {
  if ((mdl.F && !num) || (mdl.M && !cat) || (rows && !out)) return fail(COFACTOR_ERR_INVALID, "null argument");
  if (mdl.F > COFACTOR_MAX_NUM || mdl.M > COFACTOR_MAX_CAT) return fail(COFACTOR_ERR_UNSUPPORTED, "too many columns");
  if (rows == 0) return COFACTOR_OK;
  CTX_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  const size_t ncol = (size_t)mdl.F + mdl.M + 1;
  uint32_t *d = nullptr;
  HIP_TRY(hipMalloc((void **)&d, ncol * rows * 4));
  const float *dn[COFACTOR_MAX_NUM]; const int32_t *dc[COFACTOR_MAX_CAT];
  hipError_t e = hipSuccess;
  for (int i = 0; i < mdl.F && e == hipSuccess; i++) {
    dn[i] = reinterpret_cast<const float *>(d + (size_t)i * rows);
    e = num[i] ? hipMemcpyAsync((void *)dn[i], num[i], rows * 4, hipMemcpyHostToDevice, ctx->stream) : hipErrorInvalidValue;
  }
  for (int i = 0; i < mdl.M && e == hipSuccess; i++) {
    dc[i] = reinterpret_cast<const int32_t *>(d + (size_t)(mdl.F + i) * rows);
    e = cat[i] ? hipMemcpyAsync((void *)dc[i], cat[i], rows * 4, hipMemcpyHostToDevice, ctx->stream) : hipErrorInvalidValue;
  }
  cofactor_status s = COFACTOR_OK;
  uint32_t *d_out = d + (size_t)(mdl.F + mdl.M) * rows;
  if (e == hipSuccess)
    s = predict_device(ctx, mdl, argmax, emit_label, noise, seed, dn, dc, nullptr, rows,
                       argmax ? nullptr : reinterpret_cast<float *>(d_out),
                       argmax ? reinterpret_cast<int32_t *>(d_out) : nullptr);
  if (e == hipSuccess && s == COFACTOR_OK) e = hipMemcpy(out, d_out, rows * 4, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(COFACTOR_ERR_HIP, hipGetErrorString(e));
  return s;
}

cofactor_status cofactor_linreg_predict_host(cofactor_ctx *ctx, const float *params,
                                             uint64_t n_params, int32_t noise, int32_t normalize,
                                             uint64_t seed, const float *const *num,
                                             int32_t n_num, const int32_t *const *cat,
                                             int32_t n_cat, uint64_t rows, float *out) {
  if (!ctx || !params || n_num < 0 || n_cat < 0) return fail(COFACTOR_ERR_INVALID, "null argument");
  PredictModel mdl;
  std::string err;
  if (!linreg_model(params, n_params, n_num, n_cat, noise != 0, normalize != 0, mdl, err))
    return fail(COFACTOR_ERR_INVALID, err);
  return predict_host(ctx, mdl, false, false, noise != 0, seed, num, cat, rows, out);
}

cofactor_status cofactor_lda_predict_host(cofactor_ctx *ctx, const float *params,
                                          uint64_t n_params, int32_t normalize,
                                          int32_t emit_label, const float *const *num,
                                          int32_t n_num, const int32_t *const *cat, int32_t n_cat,
                                          uint64_t rows, int32_t *out) {
  if (!ctx || !params || n_num < 0 || n_cat < 0) return fail(COFACTOR_ERR_INVALID, "null argument");
  PredictModel mdl;
  std::string err;
  if (!lda_model(params, n_params, n_num, n_cat, normalize != 0, mdl, err))
    return fail(COFACTOR_ERR_INVALID, err);
  return predict_host(ctx, mdl, true, emit_label != 0, false, 0, num, cat, rows, out);
}

}  // extern "C"
