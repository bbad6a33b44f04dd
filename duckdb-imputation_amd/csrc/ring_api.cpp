// C ABI of the batched ring operations (include/cofactor_hip.h, "batched ring ops on the GPU"):
// to_cofactor, sum_triple and multiply_triple on vectors of triples, and the GROUP BY state pool.
// Kernels: ring.hip.  There is no CPU path: every entry point needs the context's GPU.
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "ring.hpp"
#include "state.hpp"

using namespace cofactor;
using namespace cofactor::detail;

namespace {

uint64_t tri64(uint64_t k) { return k * (k + 1) / 2; }

// a temporary device allocation freed on scope exit
struct DevBuf {
  void *p = nullptr;
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
  ~DevBuf() { (void)hipFree(p); }
  template <typename T> T *as() { return reinterpret_cast<T *>(p); }
};

bool tvec_dense_ok(const cofactor_tvec *v) {
  return v->N && v->lin_e && v->quad_e && (v->n == 0 || (v->lin && v->quad));
}
bool tvec_lists_ok(const cofactor_tvec *v) {
  if (v->m == 0) return true;
  if (!v->lc_outer || !v->lc_sub || !v->lc_key || !v->lc_val) return false;
  if (v->kind) return true;
  if (v->n > 0 && (!v->nc_outer || !v->nc_sub || !v->nc_key || !v->nc_val)) return false;
  return v->cc_outer && v->cc_sub && v->cc_key1 && v->cc_key2 && v->cc_val;
}

// Exclusive scan of `len` (items entries) into `offs`; *total = sum.  Synchronises the stream.
cofactor_status scan_lengths(cofactor_ctx *ctx, uint64_t *len, uint64_t *offs, uint64_t items, uint64_t *total) {
  *total = 0;
  if (items == 0) return COFACTOR_OK;
  if (items > 0x7fffffffull) return fail(COFACTOR_ERR_UNSUPPORTED, "too many sub-lists for one call");
  size_t tb = 0;
  HIP_TRY(ring_exclusive_scan(len, offs, items, nullptr, &tb, ctx->stream));
  DevBuf temp;
  HIP_TRY(temp.alloc(tb));
  HIP_TRY(ring_exclusive_scan(len, offs, items, temp.p, &tb, ctx->stream));
  uint64_t last_off = 0, last_len = 0;
  HIP_TRY(hipMemcpyAsync(&last_off, offs + items - 1, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(&last_len, len + items - 1, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  *total = last_off + last_len;
  return COFACTOR_OK;
}

}  // namespace

struct cofactor_groups {
  cofactor_ctx *ctx = nullptr;
  int n = 0, m = 0, kind = 0, is_key = 0;
  CatLayout L{};                 // key columns: dictionary geometry + table offsets inside a group's row
  CatDevice D{};                 // their dictionaries (the count / sum / pair members are not used)
  CatLayout Lg{};                // the group-key dictionary (one column), is_key only
  CatDevice Dg{};
  int32_t nkeys[COFACTOR_MAX_CAT] = {0};
  long long groups = 0, gcap = 0, dtot = 0;
  double *tab = nullptr;         // [gcap][dtot]
  bool dict_dirty = true;        // host copies below are stale
  std::vector<std::vector<int32_t>> key_of;    // [column][code] -> key
  std::vector<int32_t> group_key;              // [group row] -> key (is_key)
  std::map<int32_t, int32_t> group_of_key;
  // host batches (DataChunks of 2048 rows) are collected in pinned memory, [group id | columns][stage_cap]
  // words, and reach the device stage_cap rows at a time (groups_flush): one staging copy and one
  // update per 2^18 rows instead of per chunk — and batches big enough for the segmented path
  uint32_t *h_stage = nullptr, *d_stage = nullptr;
  uint64_t stage_cap = 0, stage_fill = 0;
};

namespace {

int dense_len(const cofactor_groups *g) { return 1 + g->n + (g->kind ? g->n : g->n * (g->n + 1) / 2); }

// allocates dictionaries for layout L (tables sized for an NB state: they are not used here)
cofactor_status dict_alloc(CatLayout L, CatDevice &D, hipStream_t st) {
  L.kind = 1;
  if (!cat_finish_layout(L)) return fail(COFACTOR_ERR_UNSUPPORTED, "dictionary too large");
  return cat_alloc(L, D, true, st);
}

// (re)shapes the group table for layout Ln and capacity cap, carrying every value over
cofactor_status groups_reshape(cofactor_groups *g, const CatLayout &Ln, long long cap) {
  hipStream_t st = g->ctx->stream;
  const long long dtn = dense_len(g) + Ln.n_cnt + Ln.n_s + Ln.n_p;
  if ((double)cap * (double)dtn * 8.0 > 200e9)
    return fail(COFACTOR_ERR_UNSUPPORTED, "GROUP BY state pool would exceed 200 GB (groups x table cells)");
  double *tn = nullptr;
  HIP_TRY(hipMalloc((void **)&tn, sizeof(double) * (size_t)cap * (size_t)dtn));
  hipError_t e = hipMemsetAsync(tn, 0, sizeof(double) * (size_t)cap * (size_t)dtn, st);
  if (e == hipSuccess && g->tab && g->groups > 0) e = launch_groups_relayout(g->L, Ln, g->tab, tn, g->dtot, dtn, g->groups, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { (void)hipFree(tn); return hip_fail(e, "groups_reshape"); }
  (void)hipFree(g->tab);
  g->tab = tn; g->gcap = cap; g->dtot = dtn;
  return COFACTOR_OK;
}

// grows one dictionary's slot capacity (all columns of Lx), re-inserting every (key, code)
cofactor_status dict_grow(cofactor_groups *g, CatLayout &Lx, CatDevice &Dx, int factor) {
  hipStream_t st = g->ctx->stream;
  CatLayout Ln = Lx;
  for (int c = 0; c < Lx.m; c++) {
    if (Ln.ht_cap[c] >= (1 << 28)) return fail(COFACTOR_ERR_UNSUPPORTED, "too many distinct keys");
    Ln.ht_cap[c] *= factor;
  }
  const int keep_kind = Ln.kind;
  Ln.kind = 1;
  if (!cat_finish_layout(Ln)) return fail(COFACTOR_ERR_UNSUPPORTED, "dictionary too large");
  CatDevice Dn;
  cofactor_status s = cat_alloc(Ln, Dn, false, st);
  if (s != COFACTOR_OK) return s;
  Dn.nkeys = Dx.nkeys; Dn.flags = Dx.flags;
  HIP_TRY(launch_cat_rehash(Lx, Dx, Ln, Dn, st));
  HIP_TRY(hipStreamSynchronize(st));
  CatDevice old = Dx;
  old.nkeys = nullptr; old.flags = nullptr;
  cat_free(old);
  Dx = Dn;
  // keep the table offsets of the true kind, take the new dictionary geometry
  for (int c = 0; c < Lx.m; c++) { Lx.ht_cap[c] = Ln.ht_cap[c]; Lx.ht_off[c] = Ln.ht_off[c]; }
  Lx.n_slots = Ln.n_slots;
  (void)keep_kind;
  return COFACTOR_OK;
}

cofactor_status groups_refresh_host(cofactor_groups *g) {
  if (!g->dict_dirty) return COFACTOR_OK;
  hipStream_t st = g->ctx->stream;
  g->key_of.assign(g->m, {});
  auto pull = [&](const CatLayout &Lx, const CatDevice &Dx, int c, std::vector<int32_t> &dst, int cap_codes) -> cofactor_status {
    std::vector<unsigned long long> slot(Lx.ht_cap[c]);
    std::vector<int32_t> code(Lx.ht_cap[c]);
    HIP_TRY(hipMemcpyAsync(slot.data(), Dx.ht_slot + Lx.ht_off[c], 8 * slot.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(code.data(), Dx.ht_code + Lx.ht_off[c], 4 * code.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    dst.assign(cap_codes, 0);
    for (size_t i = 0; i < slot.size(); i++)
      if (slot[i] != 0ull && code[i] >= 0 && code[i] < cap_codes) dst[code[i]] = (int32_t)(unsigned)(slot[i] & 0xffffffffull);
    return COFACTOR_OK;
  };
  for (int c = 0; c < g->m; c++) {
    cofactor_status s = pull(g->L, g->D, c, g->key_of[c], g->nkeys[c]);
    if (s != COFACTOR_OK) return s;
  }
  g->group_key.clear();
  g->group_of_key.clear();
  if (g->is_key) {
    cofactor_status s = pull(g->Lg, g->Dg, 0, g->group_key, (int)g->groups);
    if (s != COFACTOR_OK) return s;
    for (int r = 0; r < (int)g->groups; r++) g->group_of_key[g->group_key[r]] = r;
  }
  g->dict_dirty = false;
  return COFACTOR_OK;
}

}  // namespace

extern "C" {

cofactor_status cofactor_lift_device(cofactor_ctx *ctx, const float *const *d_num, int n_num,
                                     const int32_t *const *d_cat, int n_cat, uint64_t rows,
                                     cofactor_kind kind, cofactor_tvec *out) {
  if (!ctx || !out || n_num < 0 || n_cat < 0 || n_num > COFACTOR_MAX_NUM || n_cat > COFACTOR_MAX_CAT ||
      (n_num > 0 && !d_num) || (n_cat > 0 && !d_cat))
    return fail(COFACTOR_ERR_INVALID, "lift: bad arguments");
  if (kind != COFACTOR_TRIPLE && kind != COFACTOR_NB) return fail(COFACTOR_ERR_INVALID, "unknown kind");
  NumCols num{};
  CatCols cat{};
  for (int k = 0; k < n_num; k++) { if (!d_num[k] && rows) return fail(COFACTOR_ERR_INVALID, "null column"); num.p[k] = d_num[k]; }
  for (int c = 0; c < n_cat; c++) { if (!d_cat[c] && rows) return fail(COFACTOR_ERR_INVALID, "null column"); cat.p[c] = d_cat[c]; }
  out->count = rows; out->n = n_num; out->m = n_cat; out->kind = (int32_t)kind;
  if (rows == 0) return COFACTOR_OK;
  const uint64_t need_lc = rows * n_cat, need_nc = kind ? 0 : rows * n_num * n_cat, need_cc = kind ? 0 : rows * tri64(n_cat);
  if (!tvec_dense_ok(out) || !tvec_lists_ok(out)) return fail(COFACTOR_ERR_INVALID, "lift: an output array is null");
  if (out->lc_cap < need_lc || out->nc_cap < need_nc || out->cc_cap < need_cc)
    return fail(COFACTOR_ERR_CAPACITY, "lift: payload arrays too small");
  CTX_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  HIP_TRY(launch_lift(num, cat, n_num, n_cat, (int)kind, rows, *out, ctx->stream));
  return COFACTOR_OK;
}

cofactor_status cofactor_agg_update_tvec_device(cofactor_agg *a, const cofactor_tvec *v) {
  if (!a || !v) return fail(COFACTOR_ERR_INVALID, "null argument");
  if (v->n != a->n || v->m != a->m || v->kind != a->kind)
    return fail(COFACTOR_ERR_INVALID, "sum_triple: the vector's shape differs from the state's");
  if (v->count == 0) return COFACTOR_OK;
  if (!tvec_dense_ok(v) || !tvec_lists_ok(v)) return fail(COFACTOR_ERR_INVALID, "sum_triple: an input array is null");
  cofactor_ctx *ctx = a->ctx;
  CTX_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  cofactor_status s = stage_flush(a);
  if (s != COFACTOR_OK) return s;
  hipStream_t st = ctx->stream;
  if (!ctx->ring_red) HIP_TRY(hipMalloc((void **)&ctx->ring_red, sizeof(double) * 256));
  a->blob_cache_valid = false;
  if (a->m > 0) {
    s = cat_dictionaries_with(a, [&]() { return launch_tvec_keys(*v, a->L, a->D, 0, st); });
    if (s != COFACTOR_OK) return s;
  }
  a->dev_dirty = true;
  HIP_TRY(launch_tvec_dense(*v, ctx->ring_red, a->d_acc, a->d_kept, ctx->gram_grid, st));
  if (a->m > 0) {
    HIP_TRY(launch_tvec_keys(*v, a->L, a->D, 1, st));
    a->cat_check_pending = true;
  }
  // column pairs the state keeps as sorted lists (sum.cpp:246-260 adds into a std::map): their
  // quad_cat entries become one (packed keys, count) list per pair, merged into the pair's store
  if (a->m > 0 && !a->kind && any_sparse_pair(a->L)) {
    const int P = a->m * (a->m + 1) / 2;
    a->sparse.resize(P);
    DevBuf d_counts, d_base, d_keys, d_cnt;
    std::vector<unsigned long long> counts(P, 0), base(P, 0);
    HIP_TRY(d_counts.alloc(P * 8));
    HIP_TRY(d_base.alloc(P * 8));
    HIP_TRY(hipMemsetAsync(d_counts.p, 0, P * 8, st));
    HIP_TRY(launch_tvec_sparse(*v, a->L, d_counts.as<unsigned long long>(), nullptr, nullptr, nullptr, 0, st));
    HIP_TRY(hipMemcpyAsync(counts.data(), d_counts.p, P * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    unsigned long long total = 0;
    for (int q = 0; q < P; q++) { base[q] = total; total += counts[q]; }
    if (total) {
      HIP_TRY(d_keys.alloc(total * 8));
      HIP_TRY(d_cnt.alloc(total * 8));
      HIP_TRY(hipMemcpyAsync(d_base.p, base.data(), P * 8, hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemsetAsync(d_counts.p, 0, P * 8, st));
      HIP_TRY(launch_tvec_sparse(*v, a->L, d_counts.as<unsigned long long>(), d_base.as<unsigned long long>(),
                                 d_keys.as<unsigned long long>(), d_cnt.as<unsigned long long>(), 1, st));
      for (int q = 0; q < P; q++) {
        if (!counts[q]) continue;
        hipError_t e = sparse_merge_lists(ctx->sparse_sc, a->sparse[q], d_keys.as<unsigned long long>() + base[q],
                                          d_cnt.as<unsigned long long>() + base[q], (size_t)counts[q], st);
        if (e == hipErrorInvalidValue) return fail(COFACTOR_ERR_UNSUPPORTED, "a sparse pair table would pass 2^31 entries");
        if (e != hipSuccess) return hip_fail(e, "sum_triple: sparse pair table");
      }
      HIP_TRY(hipStreamSynchronize(st));             // the lists above are freed on return
    }
  }
  return COFACTOR_OK;
}

cofactor_status cofactor_multiply_device(cofactor_ctx *ctx, const cofactor_tvec *a, const uint32_t *d_a_sel,
                                         const cofactor_tvec *b, const uint32_t *d_b_sel, uint64_t rows,
                                         cofactor_tvec *out, uint64_t *lc_need, uint64_t *nc_need,
                                         uint64_t *cc_need) {
  if (!ctx || !a || !b) return fail(COFACTOR_ERR_INVALID, "null argument");
  if (a->kind != b->kind) return fail(COFACTOR_ERR_INVALID, "multiply: triple kinds differ");
  if (!tvec_dense_ok(a) || !tvec_lists_ok(a) || !tvec_dense_ok(b) || !tvec_lists_ok(b))
    return fail(COFACTOR_ERR_INVALID, "multiply: an input array is null");
  const int nR = a->n + b->n, mR = a->m + b->m, kind = a->kind;
  const bool fam[3] = {mR > 0, !kind && nR * mR > 0, !kind && mR > 0};
  CTX_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  hipStream_t st = ctx->stream;
  if (rows > 0x7fffffffull) return fail(COFACTOR_ERR_UNSUPPORTED, "too many rows for one call");
  // The plan of a call: per row the entries of its three list families (mulfill.hip) and their
  // exclusive scans over the ROWS, in one context scratch block (no allocation per call).  A size
  // query leaves its plan behind; the fill call that follows it with the same arguments takes it
  // over instead of computing it again (the inputs must not change in between).
  const uintptr_t key[8] = {(uintptr_t)a->N, (uintptr_t)a->lc_sub, (uintptr_t)b->N, (uintptr_t)b->lc_sub,
                            (uintptr_t)d_a_sel, (uintptr_t)d_b_sel, (uintptr_t)rows,
                            (uintptr_t)(a->n | (a->m << 8) | (b->n << 16) | (b->m << 24)) ^ ((uintptr_t)kind << 40)};
  const bool reuse = out && ctx->mul_plan_valid && std::memcmp(key, ctx->mul_plan_key, sizeof(key)) == 0;
  ctx->mul_plan_valid = false;
  uint64_t need[3] = {0, 0, 0};
  size_t temp_bytes = 0;
  if (rows) HIP_TRY(ring_exclusive_scan(nullptr, nullptr, rows, nullptr, &temp_bytes, st));
  const size_t col = (rows * 8 + 255) & ~(size_t)255;
  const size_t total_bytes = 6 * col + ((temp_bytes + 255) & ~(size_t)255) + 512;
  if (!reuse && total_bytes > ctx->ring_scratch_bytes) {
    HIP_TRY(hipStreamSynchronize(st));               // kernels of earlier calls may still read the old block
    (void)hipFree(ctx->ring_scratch);
    ctx->ring_scratch = nullptr;
    ctx->ring_scratch_bytes = 0;
    HIP_TRY(hipMalloc(&ctx->ring_scratch, total_bytes + total_bytes / 4));
    ctx->ring_scratch_bytes = total_bytes + total_bytes / 4;
  }
  char *blk = reinterpret_cast<char *>(ctx->ring_scratch);
  uint64_t *tot[3], *base[3];
  for (int f = 0; f < 3; f++) { tot[f] = reinterpret_cast<uint64_t *>(blk + f * col); base[f] = reinterpret_cast<uint64_t *>(blk + (3 + f) * col); }
  void *temp = blk + 6 * col;
  if (reuse) {
    for (int f = 0; f < 3; f++) need[f] = ctx->mul_plan_need[f];
  } else if (rows) {
    uint64_t tail[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    HIP_TRY(launch_mul_pair_lens(*a, d_a_sel, *b, d_b_sel, rows, tot[0], tot[1], tot[2], st));
    for (int f = 0; f < 3; f++) {
      if (!fam[f]) continue;
      HIP_TRY(ring_exclusive_scan(tot[f], base[f], rows, temp, &temp_bytes, st));
      HIP_TRY(hipMemcpyAsync(&tail[f][0], base[f] + rows - 1, 8, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(&tail[f][1], tot[f] + rows - 1, 8, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));               // the ONE synchronisation: the three payload sizes
    for (int f = 0; f < 3; f++) need[f] = tail[f][0] + tail[f][1];
  }
  if (lc_need) *lc_need = need[0];
  if (nc_need) *nc_need = need[1];
  if (cc_need) *cc_need = need[2];
  if (!out) {
    std::memcpy(ctx->mul_plan_key, key, sizeof(key));
    for (int f = 0; f < 3; f++) ctx->mul_plan_need[f] = need[f];
    ctx->mul_plan_valid = rows > 0;
    return COFACTOR_OK;
  }
  out->count = rows; out->n = nR; out->m = mR; out->kind = kind;
  if (rows == 0) return COFACTOR_OK;
  if (out->lc_cap < need[0] || out->nc_cap < need[1] || out->cc_cap < need[2])
    return fail(COFACTOR_ERR_CAPACITY, "multiply: payload arrays too small");
  if (!tvec_dense_ok(out) || !tvec_lists_ok(out)) return fail(COFACTOR_ERR_INVALID, "multiply: an output array is null");
  HIP_TRY(launch_mul_fill(*a, d_a_sel, *b, d_b_sel, rows, base[0], base[1], base[2], *out, need[0] + need[1] + need[2], ctx->cus, st));
  // (asynchronous from here, like every device entry point: the scratch block stays the context's)
  return COFACTOR_OK;
}

// ---- the same over host arrays ------------------------------------------------------------------------

extern "C++" {
namespace {

// Device mirror of a vector of triples: every array the shape uses, sized like the host one.
struct DevTvec {
  cofactor_tvec d{};
  std::vector<void *> owned;
  ~DevTvec() { for (void *p : owned) (void)hipFree(p); }
  template <typename T> hipError_t take(T *&dst, uint64_t count) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<uint64_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) { owned.push_back(p); dst = reinterpret_cast<T *>(p); }
    return e;
  }
  // arrays for `rows` triples of shape (n, m, kind) with the given payload capacities
  hipError_t shape(uint64_t rows, int n, int m, int kind, uint64_t lin_len, uint64_t quad_len, uint64_t lc_subs,
                   uint64_t nc_subs, uint64_t cc_subs, uint64_t lc, uint64_t nc, uint64_t cc) {
    d.count = rows; d.n = n; d.m = m; d.kind = kind;
    d.lin_len = lin_len; d.quad_len = quad_len; d.lc_subs = lc_subs; d.nc_subs = nc_subs; d.cc_subs = cc_subs;
    d.lc_cap = lc; d.nc_cap = nc; d.cc_cap = cc;
    hipError_t e = take(d.N, rows);
    if (e == hipSuccess) e = take(d.lin_e, 2 * rows);
    if (e == hipSuccess) e = take(d.quad_e, 2 * rows);
    if (e == hipSuccess) e = take(d.lin, lin_len);
    if (e == hipSuccess) e = take(d.quad, quad_len);
    if (e == hipSuccess) e = take(d.lc_outer, 2 * rows);
    if (e == hipSuccess) e = take(d.lc_sub, 2 * lc_subs);
    if (e == hipSuccess) e = take(d.lc_key, lc);
    if (e == hipSuccess) e = take(d.lc_val, lc);
    if (e == hipSuccess) e = take(d.nc_outer, 2 * rows);
    if (e == hipSuccess) e = take(d.nc_sub, 2 * nc_subs);
    if (e == hipSuccess) e = take(d.nc_key, nc);
    if (e == hipSuccess) e = take(d.nc_val, nc);
    if (e == hipSuccess) e = take(d.cc_outer, 2 * rows);
    if (e == hipSuccess) e = take(d.cc_sub, 2 * cc_subs);
    if (e == hipSuccess) e = take(d.cc_key1, cc);
    if (e == hipSuccess) e = take(d.cc_key2, cc);
    if (e == hipSuccess) e = take(d.cc_val, cc);
    return e;
  }
  // copies every array between this mirror and the host vector h (to_device or back)
  hipError_t copy(const cofactor_tvec &h, bool to_device, hipStream_t st) const {
    const hipMemcpyKind k = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
    hipError_t e = hipSuccess;
    auto mv = [&](void *dv, const void *hv, uint64_t bytes) {
      if (e != hipSuccess || bytes == 0 || !hv) return;
      e = to_device ? hipMemcpyAsync(dv, hv, bytes, k, st) : hipMemcpyAsync(const_cast<void *>(hv), dv, bytes, k, st);
    };
    const uint64_t rows = d.count;
    mv(d.N, h.N, rows * 4); mv(d.lin_e, h.lin_e, rows * 16); mv(d.quad_e, h.quad_e, rows * 16);
    mv(d.lin, h.lin, d.lin_len * 4); mv(d.quad, h.quad, d.quad_len * 4);
    if (d.m > 0) {
      mv(d.lc_outer, h.lc_outer, rows * 16); mv(d.lc_sub, h.lc_sub, d.lc_subs * 16);
      mv(d.lc_key, h.lc_key, d.lc_cap * 4); mv(d.lc_val, h.lc_val, d.lc_cap * 4);
      if (!d.kind) {
        mv(d.nc_outer, h.nc_outer, rows * 16); mv(d.nc_sub, h.nc_sub, d.nc_subs * 16);
        mv(d.nc_key, h.nc_key, d.nc_cap * 4); mv(d.nc_val, h.nc_val, d.nc_cap * 4);
        mv(d.cc_outer, h.cc_outer, rows * 16); mv(d.cc_sub, h.cc_sub, d.cc_subs * 16);
        mv(d.cc_key1, h.cc_key1, d.cc_cap * 4); mv(d.cc_key2, h.cc_key2, d.cc_cap * 4); mv(d.cc_val, h.cc_val, d.cc_cap * 4);
      }
    }
    return e;
  }
  hipError_t mirror_of(const cofactor_tvec &h) {
    return shape(h.count, h.n, h.m, h.kind, h.lin_len, h.quad_len, h.lc_subs, h.nc_subs, h.cc_subs, h.lc_cap, h.nc_cap, h.cc_cap);
  }
};

}  // namespace
}  // extern "C++"

cofactor_status cofactor_lift_host_tvec(cofactor_ctx *ctx, const float *const *num, int n_num, const int32_t *const *cat,
                                        int n_cat, uint64_t rows, cofactor_kind kind, cofactor_tvec *out) {
  if (!ctx || !out || n_num < 0 || n_cat < 0 || n_num > COFACTOR_MAX_NUM || n_cat > COFACTOR_MAX_CAT ||
      (n_num > 0 && !num) || (n_cat > 0 && !cat))
    return fail(COFACTOR_ERR_INVALID, "lift: bad arguments");
  CTX_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  const uint64_t T = kind ? (uint64_t)n_num : tri64(n_num), nm = kind ? 0 : (uint64_t)n_num * n_cat, Tm = kind ? 0 : tri64(n_cat);
  DevBuf cols;
  HIP_TRY(cols.alloc((size_t)(n_num + n_cat) * rows * 4));
  const float *dn[COFACTOR_MAX_NUM];
  const int32_t *dc[COFACTOR_MAX_CAT];
  uint32_t *d = cols.as<uint32_t>();
  for (int k = 0; k < n_num; k++) {
    if (!num[k] && rows) return fail(COFACTOR_ERR_INVALID, "null column");
    dn[k] = reinterpret_cast<const float *>(d + (size_t)k * rows);
    HIP_TRY(hipMemcpyAsync((void *)dn[k], num[k], rows * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  for (int c = 0; c < n_cat; c++) {
    if (!cat[c] && rows) return fail(COFACTOR_ERR_INVALID, "null column");
    dc[c] = reinterpret_cast<const int32_t *>(d + (size_t)(n_num + c) * rows);
    HIP_TRY(hipMemcpyAsync((void *)dc[c], cat[c], rows * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  DevTvec dv;
  HIP_TRY(dv.shape(rows, n_num, n_cat, (int)kind, rows * n_num, rows * T, rows * n_cat, rows * nm, rows * Tm, rows * n_cat,
                   rows * nm, rows * Tm));
  if (out->lc_cap < dv.d.lc_cap || out->nc_cap < dv.d.nc_cap || out->cc_cap < dv.d.cc_cap)
    return fail(COFACTOR_ERR_CAPACITY, "lift: payload arrays too small");
  cofactor_status s = cofactor_lift_device(ctx, dn, n_num, dc, n_cat, rows, kind, &dv.d);
  if (s != COFACTOR_OK) return s;
  out->count = rows; out->n = n_num; out->m = n_cat; out->kind = (int32_t)kind;
  HIP_TRY(dv.copy(*out, /*to_device=*/false, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return COFACTOR_OK;
}

// A host vector of triples before it goes to the device: every (offset, length) entry must stay inside
// the extent of the array it points into — the kernels index the child arrays with these entries
// as they are, and a NULL or malformed row that came through SQL carries uninitialised ones (an
// out-of-bounds device read is a GPU fault, not an error code).  Row shapes are checked too.
static cofactor_status validate_host_tvec(const cofactor_tvec &t, const char *what) {
  auto bad = [&](const char *part) { return fail(COFACTOR_ERR_INVALID, std::string(what) + ": " + part); };
  if (t.n < 0 || t.n > COFACTOR_MAX_NUM || t.m < 0 || t.m > COFACTOR_MAX_CAT || (t.kind != 0 && t.kind != 1))
    return bad("bad n / m / kind");
  if (t.count == 0) return COFACTOR_OK;
  const uint64_t n = (uint64_t)t.n, m = (uint64_t)t.m, T = t.kind ? n : n * (n + 1) / 2;
  if (!t.N || !t.lin_e || !t.quad_e || (n && (!t.lin || !t.quad))) return bad("null dense member");
  auto inside = [](uint64_t off, uint64_t len, uint64_t extent) { return off <= extent && len <= extent - off; };
  for (uint64_t i = 0; i < t.count; i++) {
    if (t.lin_e[2 * i + 1] != n || !inside(t.lin_e[2 * i], n, t.lin_len)) return bad("a lin entry leaves the lin array (or is not n long)");
    if (t.quad_e[2 * i + 1] != T || !inside(t.quad_e[2 * i], T, t.quad_len)) return bad("a quad entry leaves the quad array (or has the wrong length)");
  }
  auto lists = [&](const uint64_t *outer, const uint64_t *sub, uint64_t subs, uint64_t cap, uint64_t per_row,
                   const void *k1, const void *k2, const void *val, bool two, const char *name) -> cofactor_status {
    if (per_row == 0) return COFACTOR_OK;
    if (!outer || !sub) return bad("null list member");
    for (uint64_t i = 0; i < t.count; i++) {
      const uint64_t off = outer[2 * i], len = outer[2 * i + 1];
      if (len != per_row || !inside(off, len, subs)) return fail(COFACTOR_ERR_INVALID, std::string(what) + ": an outer " + name + " entry leaves its sub-list array (or has the wrong length)");
      for (uint64_t j = off; j < off + len; j++)
        if (!inside(sub[2 * j], sub[2 * j + 1], cap)) return fail(COFACTOR_ERR_INVALID, std::string(what) + ": a " + name + " sub-list leaves its payload arrays");
    }
    if (cap && (!k1 || !val || (two && !k2))) return bad("null payload member");
    return COFACTOR_OK;
  };
  cofactor_status s = lists(t.lc_outer, t.lc_sub, t.lc_subs, t.lc_cap, m, t.lc_key, nullptr, t.lc_val, false, "lin_cat");
  if (s != COFACTOR_OK || t.kind) return s;
  s = lists(t.nc_outer, t.nc_sub, t.nc_subs, t.nc_cap, n * m, t.nc_key, nullptr, t.nc_val, false, "quad_num_cat");
  if (s != COFACTOR_OK) return s;
  return lists(t.cc_outer, t.cc_sub, t.cc_subs, t.cc_cap, m * (m + 1) / 2, t.cc_key1, t.cc_key2, t.cc_val, true, "quad_cat");
}

cofactor_status cofactor_agg_update_tvec_host(cofactor_agg *a, const cofactor_tvec *v) {
  if (!a || !v) return fail(COFACTOR_ERR_INVALID, "null argument");
  if (v->count == 0) return COFACTOR_OK;
  {
    cofactor_status vs = validate_host_tvec(*v, "update_tvec_host");
    if (vs != COFACTOR_OK) return vs;
  }
  CTX_LOCK(a->ctx);
  DeviceGuard guard(a->ctx->device);
  DevTvec dv;
  HIP_TRY(dv.mirror_of(*v));
  HIP_TRY(dv.copy(*v, /*to_device=*/true, a->ctx->stream));
  cofactor_status s = cofactor_agg_update_tvec_device(a, &dv.d);
  hipError_t e = hipStreamSynchronize(a->ctx->stream);   // the mirror is freed on return
  if (s != COFACTOR_OK) return s;
  if (e != hipSuccess) return hip_fail(e, "update_tvec_host");
  return COFACTOR_OK;
}

cofactor_status cofactor_multiply_host(cofactor_ctx *ctx, const cofactor_tvec *a, const uint32_t *a_sel,
                                       const cofactor_tvec *b, const uint32_t *b_sel, uint64_t rows, cofactor_tvec *out,
                                       uint64_t *lc_need, uint64_t *nc_need, uint64_t *cc_need) {
  if (!ctx || !a || !b) return fail(COFACTOR_ERR_INVALID, "null argument");
  {
    cofactor_status vs = validate_host_tvec(*a, "multiply_host (left)");
    if (vs == COFACTOR_OK) vs = validate_host_tvec(*b, "multiply_host (right)");
    if (vs != COFACTOR_OK) return vs;
    for (uint64_t i = 0; i < rows; i++)
      if ((a_sel ? a_sel[i] : i) >= a->count || (b_sel ? b_sel[i] : i) >= b->count)
        return fail(COFACTOR_ERR_INVALID, "multiply_host: a selection entry points past its vector");
  }
  CTX_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  hipStream_t st = ctx->stream;
  DevTvec da, db;
  HIP_TRY(da.mirror_of(*a));
  HIP_TRY(db.mirror_of(*b));
  HIP_TRY(da.copy(*a, true, st));
  HIP_TRY(db.copy(*b, true, st));
  DevBuf sa, sb;
  if (a_sel) { HIP_TRY(sa.alloc(rows * 4)); HIP_TRY(hipMemcpyAsync(sa.p, a_sel, rows * 4, hipMemcpyHostToDevice, st)); }
  if (b_sel) { HIP_TRY(sb.alloc(rows * 4)); HIP_TRY(hipMemcpyAsync(sb.p, b_sel, rows * 4, hipMemcpyHostToDevice, st)); }
  uint64_t need[3] = {0, 0, 0};
  cofactor_status s = cofactor_multiply_device(ctx, &da.d, a_sel ? sa.as<uint32_t>() : nullptr, &db.d,
                                               b_sel ? sb.as<uint32_t>() : nullptr, rows, nullptr, &need[0], &need[1], &need[2]);
  if (s != COFACTOR_OK) return s;
  if (lc_need) *lc_need = need[0];
  if (nc_need) *nc_need = need[1];
  if (cc_need) *cc_need = need[2];
  if (!out) return COFACTOR_OK;
  if (out->lc_cap < need[0] || out->nc_cap < need[1] || out->cc_cap < need[2])
    return fail(COFACTOR_ERR_CAPACITY, "multiply: payload arrays too small");
  const int nR = a->n + b->n, mR = a->m + b->m, kind = a->kind;
  DevTvec dout;
  HIP_TRY(dout.shape(rows, nR, mR, kind, rows * nR, rows * (kind ? (uint64_t)nR : tri64(nR)), rows * mR,
                     kind ? 0 : rows * nR * mR, kind ? 0 : rows * tri64(mR), need[0], need[1], need[2]));
  s = cofactor_multiply_device(ctx, &da.d, a_sel ? sa.as<uint32_t>() : nullptr, &db.d, b_sel ? sb.as<uint32_t>() : nullptr,
                               rows, &dout.d, nullptr, nullptr, nullptr);
  if (s != COFACTOR_OK) return s;
  out->count = rows; out->n = nR; out->m = mR; out->kind = kind;
  HIP_TRY(dout.copy(*out, false, st));
  HIP_TRY(hipStreamSynchronize(st));
  return COFACTOR_OK;
}

// ---- GROUP BY state pool --------------------------------------------------------------------------

cofactor_status cofactor_groups_create(cofactor_ctx *ctx, int n_num, int n_cat, cofactor_kind kind, int is_key,
                                       cofactor_groups **out) {
  if (!ctx || !out) return fail(COFACTOR_ERR_INVALID, "ctx/out is null");
  *out = nullptr;
  if (n_num < 0 || n_num > COFACTOR_MAX_NUM || n_cat < 0 || n_cat > COFACTOR_MAX_CAT)
    return fail(COFACTOR_ERR_INVALID, "column counts must be in 0..20");
  if (kind != COFACTOR_TRIPLE && kind != COFACTOR_NB) return fail(COFACTOR_ERR_INVALID, "unknown kind");
  CTX_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  auto g = std::make_unique<cofactor_groups>();
  g->ctx = ctx; g->n = n_num; g->m = n_cat; g->kind = (int)kind; g->is_key = is_key != 0;
  g->L = CatLayout{};
  g->L.n = n_num; g->L.m = n_cat; g->L.kind = (int)kind;
  for (int c = 0; c < n_cat; c++) { g->L.ht_cap[c] = 64; g->L.kc[c] = 4; }
  if (!cat_finish_layout(g->L)) return fail(COFACTOR_ERR_UNSUPPORTED, "layout overflow");
  cofactor_status s = dict_alloc(g->L, g->D, ctx->stream);
  if (s != COFACTOR_OK) return s;
  g->Lg = CatLayout{};
  g->Lg.n = 0; g->Lg.m = 1; g->Lg.kind = 1; g->Lg.ht_cap[0] = 1024; g->Lg.kc[0] = 4;
  if (!cat_finish_layout(g->Lg)) return fail(COFACTOR_ERR_UNSUPPORTED, "layout overflow");
  s = dict_alloc(g->Lg, g->Dg, ctx->stream);
  if (s != COFACTOR_OK) { cat_free(g->D); return s; }
  g->groups = 0;
  g->dtot = dense_len(g.get()) + g->L.n_cnt + g->L.n_s + g->L.n_p;
  s = groups_reshape(g.get(), g->L, 1024);
  if (s != COFACTOR_OK) { cat_free(g->D); cat_free(g->Dg); return s; }
  *out = g.release();
  return COFACTOR_OK;
}

void cofactor_groups_destroy(cofactor_groups *g) {
  if (!g) return;
  CTX_LOCK(g->ctx);
  DeviceGuard guard(g->ctx->device);
  (void)hipStreamSynchronize(g->ctx->stream);
  cat_free(g->D);
  cat_free(g->Dg);
  (void)hipFree(g->tab);
  (void)hipFree(g->d_stage);
  (void)hipHostFree(g->h_stage);
  delete g;
}

extern "C++" {
namespace { cofactor_status groups_flush(cofactor_groups *g); }
}

cofactor_status cofactor_groups_update_device(cofactor_groups *g, const int32_t *d_gid, const float *const *d_num,
                                              const int32_t *const *d_cat, uint64_t rows) {
  if (!g || (rows && !d_gid) || (g->n > 0 && !d_num) || (g->m > 0 && !d_cat)) return fail(COFACTOR_ERR_INVALID, "null argument");
  if (rows == 0) return COFACTOR_OK;
  NumCols num{};
  CatCols cat{};
  for (int k = 0; k < g->n; k++) { if (!d_num[k]) return fail(COFACTOR_ERR_INVALID, "null column"); num.p[k] = d_num[k]; }
  for (int c = 0; c < g->m; c++) { if (!d_cat[c]) return fail(COFACTOR_ERR_INVALID, "null column"); cat.p[c] = d_cat[c]; }
  cofactor_ctx *ctx = g->ctx;
  CTX_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  hipStream_t st = ctx->stream;
  if (g->stage_fill) {                               // host rows collected earlier go first
    cofactor_status fs = groups_flush(g);
    if (fs != COFACTOR_OK) return fs;
  }
  // wide numeric triples with many rows per group take the segmented path (groupseg.hip)
  constexpr uint64_t SEG_MAX_ROWS = 1ull << 27;
  const bool seg_shape = g->m == 0 && g->kind == 0 && g->n >= 1;
  auto seg_wanted = [&](long long groups) {
    return seg_shape && groups > 0 && groups < (1ll << 27) && ctx->groups_seg != 2 &&
           (ctx->groups_seg == 1 || (g->n >= 2 && rows >= (1ull << 18) && rows >= 32ull * (uint64_t)groups));
  };
  auto seg_scratch = [&](uint64_t step, long long groups) -> cofactor_status {
    const size_t need = groups_seg_scratch_bytes(g->n, step, groups, ctx->cus);
    if (need > ctx->seg_scratch_bytes) {
      HIP_TRY(hipStreamSynchronize(st));
      (void)hipFree(ctx->seg_scratch);
      ctx->seg_scratch = nullptr;
      ctx->seg_scratch_bytes = 0;
      HIP_TRY(hipMalloc(&ctx->seg_scratch, need + need / 8));
      ctx->seg_scratch_bytes = need + need / 8;
    }
    return COFACTOR_OK;
  };
  // 0. key-typed groups that are all known already (every batch but the first few): the segmented
  //    path's counting pass probes the dictionary anyway, so it doubles as the check and the insert
  //    pass is skipped
  if (g->is_key && rows <= SEG_MAX_ROWS && seg_wanted(g->groups)) {
    cofactor_status s = seg_scratch(rows, g->groups);
    if (s != COFACTOR_OK) return s;
    HIP_TRY(hipMemsetAsync(g->Dg.flags + 2, 0, sizeof(int32_t), st));
    HIP_TRY(launch_groups_segmented(d_gid, num, g->n, rows, g->Lg, g->Dg, 1, g->groups, g->tab, g->dtot, ctx->seg_scratch,
                                    ctx->cus, g->Dg.flags + 2, 2, st));
    int32_t miss = 0;
    HIP_TRY(hipMemcpyAsync(&miss, g->Dg.flags + 2, sizeof(miss), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (!miss) {
      HIP_TRY(launch_groups_segmented(d_gid, num, g->n, rows, g->Lg, g->Dg, 1, g->groups, g->tab, g->dtot, ctx->seg_scratch,
                                      ctx->cus, g->Dg.flags + 2, 1, st));
      return COFACTOR_OK;
    }
  }
  g->dict_dirty = true;
  // 1. keys of the batch into the dictionaries (grown and re-run while one of them runs full)
  for (int attempt = 0;; attempt++) {
    HIP_TRY(hipMemsetAsync(g->D.flags, 0, sizeof(int32_t), st));
    HIP_TRY(hipMemsetAsync(g->Dg.flags, 0, sizeof(int32_t), st));
    HIP_TRY(launch_groups_insert(d_gid, cat, rows, g->L, g->D, g->Lg, g->Dg, g->is_key, st));
    int32_t f1[4] = {0, 0, 0, 0}, f2[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(f1, g->D.flags, sizeof(f1), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(f2, g->Dg.flags, sizeof(f2), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (f1[1] || f2[1]) return fail(COFACTOR_ERR_INTERNAL, "a row of an earlier batch met a key missing from its dictionary");
    if (!f1[0] && !f2[0]) break;
    if (attempt > 24) return fail(COFACTOR_ERR_UNSUPPORTED, "dictionary growth did not converge");
    cofactor_status s = COFACTOR_OK;
    if (f1[0]) s = dict_grow(g, g->L, g->D, 4);
    if (s == COFACTOR_OK && f2[0]) s = dict_grow(g, g->Lg, g->Dg, 4);
    if (s != COFACTOR_OK) return s;
  }
  // 2. codes for the new keys; the table grows with the code capacities and the number of groups
  HIP_TRY(launch_cat_assign_codes(g->L, g->D, st));
  if (g->is_key) HIP_TRY(launch_cat_assign_codes(g->Lg, g->Dg, st));
  int32_t nk[COFACTOR_MAX_CAT] = {0}, ng[COFACTOR_MAX_CAT] = {0};
  HIP_TRY(hipMemcpyAsync(nk, g->D.nkeys, sizeof(nk), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(ng, g->Dg.nkeys, sizeof(ng), hipMemcpyDeviceToHost, st));
  long long groups = g->groups;
  if (!g->is_key) {
    DevBuf mx;
    HIP_TRY(mx.alloc(4));
    HIP_TRY(hipMemsetAsync(mx.p, 0xFF, 4, st));       // -1
    HIP_TRY(launch_max_i32(d_gid, rows, mx.as<int>(), st));
    int top = -1;
    HIP_TRY(hipMemcpyAsync(&top, mx.p, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    groups = std::max<long long>(groups, (long long)top + 1);
  } else {
    HIP_TRY(hipStreamSynchronize(st));
    groups = ng[0];
  }
  CatLayout Ln = g->L;
  bool relayout = false;
  for (int c = 0; c < g->m; c++) {
    g->nkeys[c] = nk[c];
    if (nk[c] > Ln.kc[c]) { Ln.kc[c] = next_pow2(nk[c]); relayout = true; }
  }
  // load factor <= 1/2 for the next batches
  bool grow_d = false, grow_g = false;
  for (int c = 0; c < g->m; c++) grow_d = grow_d || nk[c] * 2 > g->L.ht_cap[c];
  grow_g = g->is_key && ng[0] * 2 > g->Lg.ht_cap[0];
  if (grow_d) { cofactor_status s = dict_grow(g, g->L, g->D, 4); if (s != COFACTOR_OK) return s; for (int c = 0; c < g->m; c++) { Ln.ht_cap[c] = g->L.ht_cap[c]; } }
  if (grow_g) { cofactor_status s = dict_grow(g, g->Lg, g->Dg, 4); if (s != COFACTOR_OK) return s; }
  long long cap = g->gcap;
  while (cap < groups) cap *= 2;
  if (relayout || cap != g->gcap) {
    if (!cat_finish_layout(Ln)) return fail(COFACTOR_ERR_UNSUPPORTED, "categorical cardinalities too high for per-group tables");
    cofactor_status s = groups_reshape(g, Ln, cap);
    if (s != COFACTOR_OK) return s;
    g->L = Ln;
  }
  g->groups = groups;
  // 3. wide numeric triples with many rows per group: regroup the rows, then one matrix-core pass per group
  //    (groupseg.hip); everything else: one launch, one atomic per cell per row
  if (seg_wanted(groups)) {
    const uint64_t step = std::min<uint64_t>(rows, SEG_MAX_ROWS);
    cofactor_status s = seg_scratch(step, groups);
    if (s != COFACTOR_OK) return s;
    for (uint64_t r0 = 0; r0 < rows; r0 += step) {
      NumCols part{};
      for (int k = 0; k < g->n; k++) part.p[k] = num.p[k] + r0;
      HIP_TRY(launch_groups_segmented(d_gid + r0, part, g->n, std::min<uint64_t>(step, rows - r0), g->Lg, g->Dg, g->is_key,
                                      groups, g->tab, g->dtot, ctx->seg_scratch, ctx->cus, g->Dg.flags + 1, 0, st));
    }
    return COFACTOR_OK;
  }
  HIP_TRY(launch_groups_accumulate(d_gid, num, cat, rows, g->L, g->D, g->Lg, g->Dg, g->is_key, g->tab, g->dtot,
                                   ctx->cus * 8, st));
  return COFACTOR_OK;
}

extern "C++" {
namespace {
// staged host rows -> device -> tables.  Called (under the context lock) by every entry point that
// looks at the pool's state.
cofactor_status groups_flush(cofactor_groups *g) {
  const uint64_t rows = g->stage_fill;
  if (rows == 0) return COFACTOR_OK;
  g->stage_fill = 0;                                  // (update_device flushes too: not twice)
  cofactor_ctx *ctx = g->ctx;
  const size_t ncol = (size_t)g->n + g->m + 1;
  const float *dn[COFACTOR_MAX_NUM];
  const int32_t *dc[COFACTOR_MAX_CAT];
  for (size_t col = 0; col < ncol; col++)
    HIP_TRY(hipMemcpyAsync(g->d_stage + col * g->stage_cap, g->h_stage + col * g->stage_cap, rows * 4, hipMemcpyHostToDevice, ctx->stream));
  for (int k = 0; k < g->n; k++) dn[k] = reinterpret_cast<const float *>(g->d_stage + (size_t)(1 + k) * g->stage_cap);
  for (int c = 0; c < g->m; c++) dc[c] = reinterpret_cast<const int32_t *>(g->d_stage + (size_t)(1 + g->n + c) * g->stage_cap);
  cofactor_status s = cofactor_groups_update_device(g, reinterpret_cast<const int32_t *>(g->d_stage), dn, dc, rows);
  hipError_t e = hipStreamSynchronize(ctx->stream);   // the pinned block is written again by the next batch
  if (s != COFACTOR_OK) return s;
  if (e != hipSuccess) return hip_fail(e, "groups_flush");
  return COFACTOR_OK;
}
}  // namespace
}

cofactor_status cofactor_groups_update_host(cofactor_groups *g, const int32_t *gid, const float *const *num,
                                            const int32_t *const *cat, uint64_t rows) {
  if (!g || (rows && !gid) || (g->n > 0 && !num) || (g->m > 0 && !cat)) return fail(COFACTOR_ERR_INVALID, "null argument");
  if (rows == 0) return COFACTOR_OK;
  for (int k = 0; k < g->n; k++) if (!num[k]) return fail(COFACTOR_ERR_INVALID, "null column");
  for (int c = 0; c < g->m; c++) if (!cat[c]) return fail(COFACTOR_ERR_INVALID, "null column");
  cofactor_ctx *ctx = g->ctx;
  CTX_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  const size_t ncol = (size_t)g->n + g->m + 1;
  if (!g->h_stage) {
    const uint64_t cap = std::max<uint64_t>(ctx->stage_rows_max, 1u << 18);
    HIP_TRY(hipHostMalloc((void **)&g->h_stage, ncol * cap * 4, hipHostMallocDefault));
    hipError_t e = hipMalloc((void **)&g->d_stage, ncol * cap * 4);
    if (e != hipSuccess) { (void)hipHostFree(g->h_stage); g->h_stage = nullptr; return hip_fail(e, "groups staging"); }
    g->stage_cap = cap;
  }
  for (uint64_t done = 0; done < rows;) {
    const uint64_t take = std::min(rows - done, g->stage_cap - g->stage_fill);
    std::memcpy(g->h_stage + g->stage_fill, gid + done, take * 4);
    for (int k = 0; k < g->n; k++) std::memcpy(g->h_stage + (size_t)(1 + k) * g->stage_cap + g->stage_fill, num[k] + done, take * 4);
    for (int c = 0; c < g->m; c++) std::memcpy(g->h_stage + (size_t)(1 + g->n + c) * g->stage_cap + g->stage_fill, cat[c] + done, take * 4);
    g->stage_fill += take;
    done += take;
    if (g->stage_fill == g->stage_cap) {
      cofactor_status s = groups_flush(g);
      if (s != COFACTOR_OK) return s;
    }
  }
  return COFACTOR_OK;
}

cofactor_status cofactor_groups_count(cofactor_groups *g, uint64_t *n_groups) {
  if (!g || !n_groups) return fail(COFACTOR_ERR_INVALID, "null argument");
  {
    CTX_LOCK(g->ctx);
    DeviceGuard guard(g->ctx->device);
    cofactor_status s = groups_flush(g);
    if (s != COFACTOR_OK) return s;
  }
  *n_groups = (uint64_t)g->groups;
  return COFACTOR_OK;
}

namespace {
// group id as the caller knows it (slot id or key) -> table row; -1 if there is no such group
cofactor_status group_row(cofactor_groups *g, int32_t gid, long long *row) {
  *row = -1;
  if (!g->is_key) { if (gid >= 0 && gid < g->groups) *row = gid; return COFACTOR_OK; }
  cofactor_status s = groups_refresh_host(g);
  if (s != COFACTOR_OK) return s;
  auto it = g->group_of_key.find(gid);
  if (it != g->group_of_key.end()) *row = it->second;
  return COFACTOR_OK;
}
}  // namespace

cofactor_status cofactor_groups_combine(cofactor_groups *g, int32_t dst_gid, int32_t src_gid) {
  if (!g) return fail(COFACTOR_ERR_INVALID, "null argument");
  CTX_LOCK(g->ctx);
  DeviceGuard guard(g->ctx->device);
  long long dst, src;
  cofactor_status s = groups_flush(g);
  if (s != COFACTOR_OK) return s;
  s = group_row(g, dst_gid, &dst);
  if (s == COFACTOR_OK) s = group_row(g, src_gid, &src);
  if (s != COFACTOR_OK) return s;
  if (dst < 0 || src < 0 || dst == src) return fail(COFACTOR_ERR_INVALID, "combine: no such group (or a group with itself)");
  HIP_TRY(launch_groups_combine(g->tab, g->dtot, dst, src, g->ctx->stream));
  return COFACTOR_OK;
}

cofactor_status cofactor_groups_reset_group(cofactor_groups *g, int32_t gid) {
  if (!g) return fail(COFACTOR_ERR_INVALID, "null argument");
  if (g->is_key || gid < 0) return fail(COFACTOR_ERR_INVALID, "reset_group: slot-id pools only, slot >= 0");
  CTX_LOCK(g->ctx);
  DeviceGuard guard(g->ctx->device);
  cofactor_status s = groups_flush(g);               // rows handed over for the slot's previous owner go in first
  if (s != COFACTOR_OK) return s;
  if (gid >= g->groups) return COFACTOR_OK;          // (no row yet: nothing to clear)
  HIP_TRY(hipMemsetAsync(g->tab + (long long)gid * g->dtot, 0, sizeof(double) * (size_t)g->dtot, g->ctx->stream));
  return COFACTOR_OK;
}

cofactor_status cofactor_groups_finalize(cofactor_groups *g, int32_t gid, double *out, uint64_t cap, uint64_t *needed) {
  if (!g) return fail(COFACTOR_ERR_INVALID, "null argument");
  CTX_LOCK(g->ctx);
  DeviceGuard guard(g->ctx->device);
  cofactor_status s = groups_flush(g);
  if (s == COFACTOR_OK) s = groups_refresh_host(g);
  if (s != COFACTOR_OK) return s;
  long long rowi;
  s = group_row(g, gid, &rowi);
  if (s != COFACTOR_OK) return s;
  if (rowi < 0) return fail(COFACTOR_ERR_INVALID, "finalize: no such group");
  std::vector<double> row((size_t)g->dtot);
  hipStream_t st = g->ctx->stream;
  HIP_TRY(hipMemcpyAsync(row.data(), g->tab + rowi * g->dtot, sizeof(double) * row.size(), hipMemcpyDeviceToHost, st));
  int32_t flags[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(flags, g->D.flags, sizeof(flags), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (flags[1]) return fail(COFACTOR_ERR_INTERNAL, "a row met a key missing from its dictionary");
  const CatLayout &L = g->L;
  const int n = g->n, m = g->m, Dd = dense_len(g);
  HostTriple t;
  t.shape(g->kind, n, m);
  t.N = row[0];
  for (int k = 0; k < n; k++) t.lin[k] = row[1 + k];
  for (size_t q = 0; q < t.quad.size(); q++) t.quad[q] = row[1 + n + q];
  for (int c = 0; c < m; c++)
    for (int code = 0; code < g->nkeys[c]; code++) {
      const double cnt = row[Dd + L.cnt_off[c] + code];
      if (cnt == 0.0) continue;
      auto &vals = t.col[c][g->key_of[c][code]];
      vals.assign(g->kind ? 1 : (size_t)n + 1, 0.0);
      vals[0] = cnt;
      if (!g->kind)
        for (int k = 0; k < n; k++) vals[k + 1] = row[(size_t)Dd + L.n_cnt + L.s_off[c] + (size_t)code * n + k];
    }
  if (!g->kind) {
    int q = 0;
    for (int c1 = 0; c1 < m; c1++)
      for (int c2 = c1; c2 < m; c2++, q++)
        for (int k1 = 0; k1 < g->nkeys[c1]; k1++)
          for (int k2 = 0; k2 < g->nkeys[c2]; k2++) {
            const double v = row[(size_t)Dd + L.n_cnt + L.n_s + L.p_off[q] + (size_t)k1 * L.kc[c2] + k2];
            if (v != 0.0) t.pair[q][{g->key_of[c1][k1], g->key_of[c2][k2]}] = v;
          }
  }
  std::vector<double> blob;
  t.encode(blob);
  return emit_blob(blob, out, cap, needed);
}

cofactor_status cofactor_groups_to_tvec(cofactor_groups *g, cofactor_tvec *out, int32_t *d_group_keys,
                                        uint64_t *lc_need, uint64_t *nc_need, uint64_t *cc_need) {
  if (!g) return fail(COFACTOR_ERR_INVALID, "null argument");
  cofactor_ctx *ctx = g->ctx;
  CTX_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  hipStream_t st = ctx->stream;
  cofactor_status s = groups_flush(g);
  if (s == COFACTOR_OK) s = groups_refresh_host(g);
  if (s != COFACTOR_OK) return s;
  const CatLayout &L = g->L;
  const int n = g->n, m = g->m, kind = g->kind;
  const long long G = g->groups;
  // group order (ascending key, or slot order) and, per key column, the codes in ascending key order
  std::vector<int32_t> gorder((size_t)std::max<long long>(G, 1)), gkeys((size_t)std::max<long long>(G, 1));
  for (long long r = 0; r < G; r++) gorder[r] = (int32_t)r;
  if (g->is_key) {
    std::sort(gorder.begin(), gorder.begin() + G, [&](int32_t x, int32_t y) { return g->group_key[x] < g->group_key[y]; });
    for (long long r = 0; r < G; r++) gkeys[r] = g->group_key[gorder[r]];
  } else {
    for (long long r = 0; r < G; r++) gkeys[r] = (int32_t)r;
  }
  std::vector<int32_t> ord((size_t)std::max(L.n_cnt, 1), -1), keyof((size_t)std::max(L.n_cnt, 1), 0);
  for (int c = 0; c < m; c++) {
    std::vector<int32_t> codes(g->nkeys[c]);
    for (int k = 0; k < g->nkeys[c]; k++) { codes[k] = k; keyof[L.cnt_off[c] + k] = g->key_of[c][k]; }
    std::sort(codes.begin(), codes.end(), [&](int32_t x, int32_t y) { return g->key_of[c][x] < g->key_of[c][y]; });
    for (int r = 0; r < g->nkeys[c]; r++) ord[L.cnt_off[c] + r] = codes[r];
  }
  DevBuf d_gorder, d_ord, d_keyof;
  HIP_TRY(d_gorder.alloc(gorder.size() * 4));
  HIP_TRY(d_ord.alloc(ord.size() * 4));
  HIP_TRY(d_keyof.alloc(keyof.size() * 4));
  HIP_TRY(hipMemcpyAsync(d_gorder.p, gorder.data(), gorder.size() * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_ord.p, ord.data(), ord.size() * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_keyof.p, keyof.data(), keyof.size() * 4, hipMemcpyHostToDevice, st));
  const uint64_t per[3] = {(uint64_t)m, kind ? 0 : (uint64_t)n * m, kind ? 0 : tri64(m)};
  DevBuf len[3], offs[3];
  uint64_t need[3] = {0, 0, 0};
  cofactor_tvec none{};
  for (int f = 0; f < 3; f++) {
    const uint64_t items = (uint64_t)G * per[f];
    if (items == 0) continue;
    HIP_TRY(len[f].alloc(items * 8));
    HIP_TRY(offs[f].alloc(items * 8));
    HIP_TRY(launch_groups_lists(g->tab, g->dtot, L, d_gorder.as<int32_t>(), G, d_ord.as<int32_t>(), d_keyof.as<int32_t>(), f,
                                len[f].as<uint64_t>(), nullptr, none, 0, st));
    s = scan_lengths(ctx, len[f].as<uint64_t>(), offs[f].as<uint64_t>(), items, &need[f]);
    if (s != COFACTOR_OK) return s;
  }
  if (lc_need) *lc_need = need[0];
  if (nc_need) *nc_need = need[1];
  if (cc_need) *cc_need = need[2];
  HIP_TRY(hipStreamSynchronize(st));
  if (!out) return COFACTOR_OK;
  out->count = (uint64_t)G; out->n = n; out->m = m; out->kind = kind;
  if (G == 0) return COFACTOR_OK;
  if (out->lc_cap < need[0] || out->nc_cap < need[1] || out->cc_cap < need[2])
    return fail(COFACTOR_ERR_CAPACITY, "to_tvec: payload arrays too small");
  if (!tvec_dense_ok(out) || !tvec_lists_ok(out)) return fail(COFACTOR_ERR_INVALID, "to_tvec: an output array is null");
  HIP_TRY(launch_groups_dense(g->tab, g->dtot, n, kind ? n : n * (n + 1) / 2, d_gorder.as<int32_t>(), G, *out, st));
  for (int f = 0; f < 3; f++)
    if ((uint64_t)G * per[f])
      HIP_TRY(launch_groups_lists(g->tab, g->dtot, L, d_gorder.as<int32_t>(), G, d_ord.as<int32_t>(), d_keyof.as<int32_t>(), f,
                                  nullptr, offs[f].as<uint64_t>(), *out, 1, st));
  if (d_group_keys) HIP_TRY(hipMemcpyAsync(d_group_keys, gkeys.data(), (size_t)G * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));                 // temporaries are freed on return
  return COFACTOR_OK;
}

}  // extern "C"
