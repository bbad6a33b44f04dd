// DuckDB (v0.9.2 C++ extension API) glue for libcofactor_hip: registers the same SQL functions,
// argument types and STRUCT return types as the reference's ring code
// (reference: duckdb_extension/src/duckdb_imputation_extension.cpp:48-180) and forwards every
// callback across the C ABI of include/cofactor_hip.h.
//
//   sum_to_triple_<x>_<y>(FLOAT*x, INTEGER*y)   x,y in 0..20      -> update_host / combine / finalize; with GROUP BY
//                                                                     one cofactor_groups pool per query, one
//                                                                     cofactor_groups_update_host per chunk
//   sum_to_nb_agg_<x>_<y>(...)                                     -> same, kind = NB
//   sum_triple(triple), sum_nb_agg(triple)                          -> cofactor_agg_update_tvec_host (child arrays as is)
//   to_cofactor(cols...), to_nb_agg(cols...)                        -> cofactor_lift_host_tvec (writes the result's child arrays)
//   multiply_triple(a, b), multiply_nb_agg(a, b)                    -> cofactor_multiply_host (child arrays in, child arrays out)
//   linreg_train(triple, label, step, lambda, iters, variance, normalize)  -> cofactor_linreg_train
//   lda_train(triple, label, shrinkage, normalize)                  -> cofactor_lda_train
//   linreg_predict(params, noise, normalize, cols...)               -> cofactor_linreg_predict_host
//   lda_predict(params, normalize, cols...)                         -> cofactor_lda_predict_host
//
// This translation unit needs DuckDB's headers (<duckdb.hpp>, v0.9.2 @ 3c695d7b) and is built
// inside a DuckDB checkout exactly like the reference (see ../CMakeLists.txt); it is NOT compiled
// by __graft_entry__.build() because the image has no DuckDB.  All arithmetic lives below the C
// ABI; this file only moves DataChunk column pointers and nested result vectors.
#define DUCKDB_EXTENSION_MAIN

#include "duckdb_imputation_extension.hpp"

#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <random>
#include <unordered_map>

#include "cofactor_hip.h"
#include "duckdb.hpp"
#include "duckdb/function/aggregate_function.hpp"
#include "duckdb/function/scalar/nested_functions.hpp"
#include "duckdb/function/scalar_function.hpp"
#include "duckdb/main/extension_util.hpp"

namespace duckdb {
namespace cofactor_glue {

// ---- one GPU context per visible GPU ---------------------------------------------------------------
// DuckDB runs the aggregate on thread-local states and merges them with combine
// (reference: duckdb_extension/src/triple/sum/sum_state.cpp:10-114, registered at
// duckdb_imputation_extension.cpp:97-106): that seam is the multi-GPU seam of the extension.  Every
// worker thread sticks to one context (round-robin over the GPUs), the states it creates live on
// that GPU, its chunks are staged to that GPU, and cofactor_agg_combine merges states of different
// GPUs on the device (peer copy of the aligned table image over xGMI + an add kernel).
//   COFACTOR_DEVICES="0,1,2,3"   the GPUs to use (a device may be listed twice: two contexts on it)
//   COFACTOR_DEVICE=k            one GPU (round 1/2 behaviour)
//   neither                      all GPUs HIP shows
static std::vector<cofactor_ctx *> &Contexts() {
  static std::once_flag once;
  static std::vector<cofactor_ctx *> ctxs;
  std::call_once(once, [] {
    std::vector<int> devs;
    if (const char *list = std::getenv("COFACTOR_DEVICES")) {
      for (const char *p = list; *p;) {
        char *end = nullptr;
        const long v = std::strtol(p, &end, 10);
        if (end == p) break;
        devs.push_back((int)v);
        p = *end == ',' ? end + 1 : end;
      }
    } else if (const char *dev = std::getenv("COFACTOR_DEVICE")) {
      devs.push_back(std::atoi(dev));
    } else {
      const int n = cofactor_device_count();
      for (int d = 0; d < n; d++) devs.push_back(d);
    }
    if (devs.empty()) devs.push_back(0);             // (no GPU: the create below reports it)
    for (int d : devs) {
      cofactor_ctx *ctx = nullptr;
      if (cofactor_ctx_create(d, &ctx) != COFACTOR_OK)
        throw IOException("duckdb_imputation (MI355X): %s", cofactor_last_error());
      ctxs.push_back(ctx);
    }
  });
  return ctxs;
}
// the calling worker thread's context
static cofactor_ctx *Context() {
  static std::atomic<unsigned> next{0};
  auto &ctxs = Contexts();
  thread_local unsigned mine = next.fetch_add(1);
  return ctxs[mine % ctxs.size()];
}
// the GROUP BY pool of a query lives on ONE GPU (its table rows are updated by one kernel per chunk)
static cofactor_ctx *PoolContext() { return Contexts()[0]; }

static void Check(cofactor_status st) {
  if (st != COFACTOR_OK) throw InvalidInputException("duckdb_imputation (MI355X): %s", cofactor_last_error());
}

// ---- aggregate state: a POD handle, memcpy-relocatable like Triple::SumState ---------------------
// (reference: duckdb_extension/src/include/triple/sum/sum_state.h:14-57)
// A state fed by chunks that are all its own (no GROUP BY, or a group that fills whole chunks) owns
// a cofactor_agg; a state first met in a chunk with several states is a slot of the query's GROUP BY
// pool (one device table row per group, one kernel launch per chunk).  A state may hold both.
struct GroupPool {
  std::mutex mu;
  cofactor_groups *grp = nullptr;
  int32_t next_slot = 0;
  std::vector<int32_t> free_slots;                    // slots of destroyed states, cleared when handed out again
  ~GroupPool() { cofactor_groups_destroy(grp); }
};

struct RingState {
  cofactor_agg *agg;
  GroupPool *pool;
  int32_t slot;
};

// bind data of the aggregates: the return type (as the reference's VariableReturnBindData) + the pool
struct RingBindData : public VariableReturnBindData {
  explicit RingBindData(LogicalType t) : VariableReturnBindData(std::move(t)), pool(std::make_shared<GroupPool>()) {}
  std::shared_ptr<GroupPool> pool;
  // DuckDB copies bound aggregates (optimizer rewrites, CTE / view inlining, DISTINCT): the base
  // class's Copy() would hand back a plain VariableReturnBindData and the Cast<RingBindData>() in
  // RingUpdate would read a pool that is not there.  A copy SHARES the pool (it belongs to the same
  // query; states carry their pool pointer, RingCombine checks it).
  unique_ptr<FunctionData> Copy() const override {
    auto copy = make_uniq<RingBindData>(stype);
    copy->pool = pool;
    return std::move(copy);
  }
  bool Equals(const FunctionData &other_p) const override {
    auto &other = other_p.Cast<RingBindData>();
    return stype == other.stype && pool == other.pool;
  }
};

struct RingStateFunction {
  template <class STATE>
  static void Initialize(STATE &state) { state.agg = nullptr; state.pool = nullptr; state.slot = -1; }
  template <class STATE>
  static void Destroy(STATE &state, AggregateInputData &aid) {
    cofactor_agg_destroy(state.agg);
    state.agg = nullptr;
    // the slot goes back to its pool (reached through the bind data, which outlives the states): a
    // prepared statement that runs again, or thread-local states that were combined away, would
    // otherwise only ever add rows to the pool's table
    if (state.slot >= 0 && aid.bind_data) {
      auto &pool = aid.bind_data->template Cast<RingBindData>().pool;
      if (pool && pool.get() == state.pool) {
        std::lock_guard<std::mutex> lock(pool->mu);
        pool->free_slots.push_back(state.slot);
      }
    }
    state.slot = -1;
    state.pool = nullptr;
  }
  static bool IgnoreNull() { return false; }
};

// Flat vectors all the way down (reference: duckdb_extension/src/utils.cpp:3-19).
static void RecursiveFlatten(Vector &v, idx_t count) {
  v.Flatten(count);
  switch (v.GetType().InternalType()) {
  case PhysicalType::LIST: {
    auto &child = ListVector::GetEntry(v);
    RecursiveFlatten(child, ListVector::GetListSize(v));
    break;
  }
  case PhysicalType::STRUCT:
    for (auto &child : StructVector::GetEntries(v)) RecursiveFlatten(*child, count);
    break;
  default:
    break;
  }
}

static bool IsNumeric(const LogicalType &t) {       // sum_no_lift.cpp:69
  return t == LogicalType::FLOAT || t == LogicalType::DOUBLE;
}

// ---- result STRUCT types -----------------------------------------------------------------------
static LogicalType TripleType(bool nb, bool aggregate_names) {   // sum_no_lift.cpp:14-49, lift.cpp:246-284
  child_list_t<LogicalType> kv, kkv, fields;
  kv.emplace_back("key", LogicalType::INTEGER);
  kv.emplace_back("value", LogicalType::FLOAT);
  kkv.emplace_back("key1", LogicalType::INTEGER);
  kkv.emplace_back("key2", LogicalType::INTEGER);
  kkv.emplace_back("value", LogicalType::FLOAT);
  fields.emplace_back("N", LogicalType::INTEGER);
  fields.emplace_back(aggregate_names ? "lin_agg" : "lin_num", LogicalType::LIST(LogicalType::FLOAT));
  fields.emplace_back(aggregate_names ? "quad_agg" : "quad_num", LogicalType::LIST(LogicalType::FLOAT));
  fields.emplace_back("lin_cat", LogicalType::LIST(LogicalType::LIST(LogicalType::STRUCT(kv))));
  if (!nb) {
    fields.emplace_back("quad_num_cat", LogicalType::LIST(LogicalType::LIST(LogicalType::STRUCT(kv))));
    fields.emplace_back("quad_cat", LogicalType::LIST(LogicalType::LIST(LogicalType::STRUCT(kkv))));
  }
  return LogicalType::STRUCT(fields);
}

// ---- flat blob  <->  nested vectors --------------------------------------------------------------
// Appends the blob as row `row` of the STRUCT vector `result` (the order sum_state.cpp:116-464
// fills the children in).  `cursor[i]` = entries already written to child list i.
struct ListCursor {
  idx_t lin = 0, quad = 0, lc_outer = 0, lc_inner = 0, nc_outer = 0, nc_inner = 0, cc_outer = 0, cc_inner = 0;
};

static void ReserveList(Vector &list, idx_t upto) {
  ListVector::Reserve(list, upto);
  ListVector::SetListSize(list, upto);
}

static void WriteKeyValueLists(Vector &outer, idx_t row, const double *&p, idx_t lists, idx_t &outer_pos,
                               idx_t &inner_pos, bool two_keys) {
  // outer: LIST(LIST(STRUCT)).  First pass sizes, then fill (Reserve may move buffers).
  const double *q = p;
  idx_t total = 0;
  for (idx_t l = 0; l < lists; l++) { idx_t len = (idx_t)*q; q += 1 + len * (two_keys ? 3 : 2); total += len; }
  ReserveList(outer, outer_pos + lists);
  Vector &mid = ListVector::GetEntry(outer);
  ReserveList(mid, inner_pos + total);
  auto outer_entries = FlatVector::GetData<list_entry_t>(outer);
  auto mid_entries = FlatVector::GetData<list_entry_t>(mid);
  auto &fields = StructVector::GetEntries(ListVector::GetEntry(mid));
  auto k1 = FlatVector::GetData<int32_t>(*fields[0]);
  auto k2 = two_keys ? FlatVector::GetData<int32_t>(*fields[1]) : nullptr;
  auto val = FlatVector::GetData<float>(*fields[two_keys ? 2 : 1]);
  outer_entries[row].offset = outer_pos;
  outer_entries[row].length = lists;
  for (idx_t l = 0; l < lists; l++) {
    idx_t len = (idx_t)*p++;
    mid_entries[outer_pos + l].offset = inner_pos;
    mid_entries[outer_pos + l].length = len;
    for (idx_t e = 0; e < len; e++) {
      k1[inner_pos] = (int32_t)*p++;
      if (two_keys) k2[inner_pos] = (int32_t)*p++;
      val[inner_pos] = (float)*p++;                 // DuckDB FLOAT = the double rounded once
      inner_pos++;
    }
  }
  outer_pos += lists;
}

static void BlobToRow(const double *blob, Vector &result, idx_t row, ListCursor &cur) {
  auto &children = StructVector::GetEntries(result);
  const bool nb = blob[0] != 0;
  const idx_t n = (idx_t)blob[1], m = (idx_t)blob[2];
  const idx_t qn = nb ? n : n * (n + 1) / 2;
  FlatVector::GetData<int32_t>(*children[0])[row] = (int32_t)blob[3];
  const double *p = blob + 4;
  ReserveList(*children[1], cur.lin + n);
  auto lin = FlatVector::GetData<float>(ListVector::GetEntry(*children[1]));
  for (idx_t k = 0; k < n; k++) lin[cur.lin + k] = (float)*p++;
  FlatVector::GetData<list_entry_t>(*children[1])[row] = {cur.lin, n};
  cur.lin += n;
  ReserveList(*children[2], cur.quad + qn);
  auto quad = FlatVector::GetData<float>(ListVector::GetEntry(*children[2]));
  for (idx_t k = 0; k < qn; k++) quad[cur.quad + k] = (float)*p++;
  FlatVector::GetData<list_entry_t>(*children[2])[row] = {cur.quad, qn};
  cur.quad += qn;
  WriteKeyValueLists(*children[3], row, p, m, cur.lc_outer, cur.lc_inner, false);
  if (!nb) {
    WriteKeyValueLists(*children[4], row, p, n * m, cur.nc_outer, cur.nc_inner, false);
    WriteKeyValueLists(*children[5], row, p, m * (m + 1) / 2, cur.cc_outer, cur.cc_inner, true);
  }
}

// Row `row` of a (recursively flattened) triple STRUCT vector -> blob.
static void RowToBlob(Vector &triple, idx_t row, bool nb, std::vector<double> &blob) {
  auto &children = StructVector::GetEntries(triple);
  auto lin_e = FlatVector::GetData<list_entry_t>(*children[1])[row];
  auto quad_e = FlatVector::GetData<list_entry_t>(*children[2])[row];
  auto lc_e = FlatVector::GetData<list_entry_t>(*children[3])[row];
  const idx_t n = lin_e.length, m = lc_e.length;
  // a triple STRUCT that went through SQL may carry lists that do not fit its own n and m: such a
  // row has no blob (the reference reads past its lists here, sum.cpp:99-106, mul.cpp:57-69)
  bool consistent = quad_e.length == (nb ? n : n * (n + 1) / 2);
  if (!nb) {
    consistent = consistent && FlatVector::GetData<list_entry_t>(*children[4])[row].length == n * m &&
                 FlatVector::GetData<list_entry_t>(*children[5])[row].length == m * (m + 1) / 2;
  }
  if (!consistent) throw InvalidInputException("triple argument: list lengths do not match its lin_agg / lin_cat lengths");
  blob.push_back(nb ? 1 : 0); blob.push_back((double)n); blob.push_back((double)m);
  blob.push_back((double)FlatVector::GetData<int32_t>(*children[0])[row]);
  auto lin = FlatVector::GetData<float>(ListVector::GetEntry(*children[1]));
  for (idx_t k = 0; k < n; k++) blob.push_back(lin[lin_e.offset + k]);
  auto quad = FlatVector::GetData<float>(ListVector::GetEntry(*children[2]));
  for (idx_t k = 0; k < quad_e.length; k++) blob.push_back(quad[quad_e.offset + k]);
  auto lists = [&](Vector &outer, bool two_keys) {
    auto oe = FlatVector::GetData<list_entry_t>(outer)[row];
    Vector &mid = ListVector::GetEntry(outer);
    auto me = FlatVector::GetData<list_entry_t>(mid);
    auto &fields = StructVector::GetEntries(ListVector::GetEntry(mid));
    auto k1 = FlatVector::GetData<int32_t>(*fields[0]);
    auto k2 = two_keys ? FlatVector::GetData<int32_t>(*fields[1]) : nullptr;
    auto val = FlatVector::GetData<float>(*fields[two_keys ? 2 : 1]);
    for (idx_t l = 0; l < oe.length; l++) {
      auto e = me[oe.offset + l];
      blob.push_back((double)e.length);
      for (idx_t i = 0; i < e.length; i++) {
        blob.push_back(k1[e.offset + i]);
        if (two_keys) blob.push_back(k2[e.offset + i]);
        blob.push_back(val[e.offset + i]);
      }
    }
  };
  lists(*children[3], false);
  if (!nb) { lists(*children[4], false); lists(*children[5], true); }
}

// ---- aggregate callbacks --------------------------------------------------------------------------
// update: Triple::SumNoLift / Triple::sum_to_nb_agg (sum_no_lift.cpp:53-216, sum_to_nb_agg.cpp:39-146)
template <bool NB>
static void RingUpdate(Vector inputs[], AggregateInputData &aggr, idx_t cols, Vector &state_vector, idx_t count) {
  UnifiedVectorFormat sdata;
  state_vector.ToUnifiedFormat(count, sdata);
  auto states = (RingState **)sdata.data;

  vector<UnifiedVectorFormat> fmt(cols);
  vector<const float *> num;
  vector<const int32_t *> cat;
  vector<const uint32_t *> num_sel, cat_sel;
  for (idx_t j = 0; j < cols; j++) {
    inputs[j].ToUnifiedFormat(count, fmt[j]);
    const uint32_t *sel = fmt[j].sel->data();        // nullptr for a flat vector
    if (IsNumeric(inputs[j].GetType())) { num.push_back((const float *)fmt[j].data); num_sel.push_back(sel); }
    else { cat.push_back((const int32_t *)fmt[j].data); cat_sel.push_back(sel); }
  }
  RingState *first = states[sdata.sel->get_index(0)];
  bool single = true;
  for (idx_t i = 1; i < count && single; i++) single = states[sdata.sel->get_index(i)] == first;
  const cofactor_kind kind = NB ? COFACTOR_NB : COFACTOR_TRIPLE;
  auto feed = [&](RingState *st, const uint32_t *row_idx, idx_t rows) {
    if (!st->agg) Check(cofactor_agg_create(Context(), (int)num.size(), (int)cat.size(), kind, &st->agg));
    Check(cofactor_agg_update_host(st->agg, num.data(), cat.data(), num_sel.data(), cat_sel.data(), row_idx, rows));
  };
  if (single && first->slot < 0) { feed(first, nullptr, count); return; }
  // GROUP BY: the rows of states that own an aggregate go to it; all others are slots of the pool
  // and go to the device in ONE call with their slot as the per-row group id
  // (the per-row state pointers of sum_no_lift.cpp:84,94,139)
  GroupPool *pool = aggr.bind_data->template Cast<RingBindData>().pool.get();
  std::unordered_map<RingState *, vector<uint32_t>> rows_of;
  vector<int32_t> gid;
  vector<uint32_t> pooled;
  {
    std::lock_guard<std::mutex> lock(pool->mu);
    if (!pool->grp) Check(cofactor_groups_create(PoolContext(), (int)num.size(), (int)cat.size(), kind, /*is_key=*/0, &pool->grp));
    for (idx_t i = 0; i < count; i++) {
      RingState *st = states[sdata.sel->get_index(i)];
      if (st->agg && st->slot < 0) { rows_of[st].push_back((uint32_t)i); continue; }
      if (st->slot < 0) {
        if (!pool->free_slots.empty()) {              // a destroyed state's slot, cleared first
          st->slot = pool->free_slots.back();
          pool->free_slots.pop_back();
          Check(cofactor_groups_reset_group(pool->grp, st->slot));
        } else {
          st->slot = pool->next_slot++;
        }
        st->pool = pool;
      }
      if (st->pool != pool) throw InvalidInputException("duckdb_imputation: aggregate state used with a foreign group pool");
      gid.push_back(st->slot);
      pooled.push_back((uint32_t)i);
    }
  }
  for (auto &kv : rows_of) feed(kv.first, kv.second.data(), kv.second.size());
  if (pooled.empty()) return;
  vector<vector<float>> gnum(num.size(), vector<float>(pooled.size()));
  vector<vector<int32_t>> gcat(cat.size(), vector<int32_t>(pooled.size()));
  vector<const float *> pn;
  vector<const int32_t *> pc;
  for (size_t k = 0; k < num.size(); k++) {
    for (size_t i = 0; i < pooled.size(); i++) gnum[k][i] = num[k][num_sel[k] ? num_sel[k][pooled[i]] : pooled[i]];
    pn.push_back(gnum[k].data());
  }
  for (size_t c = 0; c < cat.size(); c++) {
    for (size_t i = 0; i < pooled.size(); i++) gcat[c][i] = cat[c][cat_sel[c] ? cat_sel[c][pooled[i]] : pooled[i]];
    pc.push_back(gcat[c].data());
  }
  std::lock_guard<std::mutex> lock(pool->mu);
  Check(cofactor_groups_update_host(pool->grp, gid.data(), pn.data(), pc.data(), pooled.size()));
}

// A (recursively flattened) triple STRUCT vector as a cofactor_tvec: pointers to its child arrays,
// nothing is copied (list_entry_t = {uint64 offset, uint64 length}).  Rows must share n and m
// and carry list counts that fit them, as the reference assumes (sum.cpp:99-106, mul.cpp:57-69).
static void VectorAsTvec(Vector &triple, idx_t count, bool nb, cofactor_tvec &tv) {
  auto &children = StructVector::GetEntries(triple);
  auto entries = [](Vector &list) { return reinterpret_cast<uint64_t *>(FlatVector::GetData<list_entry_t>(list)); };
  std::memset(&tv, 0, sizeof(tv));
  tv.count = count;
  tv.kind = nb ? COFACTOR_NB : COFACTOR_TRIPLE;
  tv.N = FlatVector::GetData<int32_t>(*children[0]);
  tv.lin_e = entries(*children[1]);
  tv.lin = FlatVector::GetData<float>(ListVector::GetEntry(*children[1]));
  tv.lin_len = ListVector::GetListSize(*children[1]);
  tv.quad_e = entries(*children[2]);
  tv.quad = FlatVector::GetData<float>(ListVector::GetEntry(*children[2]));
  tv.quad_len = ListVector::GetListSize(*children[2]);
  auto lists = [&](Vector &outer, bool two_keys, uint64_t *&o_e, uint64_t *&s_e, int32_t *&k1, int32_t *&k2, float *&val,
                   uint64_t &subs, uint64_t &payload) {
    o_e = entries(outer);
    Vector &mid = ListVector::GetEntry(outer);
    s_e = entries(mid);
    subs = ListVector::GetListSize(outer);
    payload = ListVector::GetListSize(mid);
    auto &fields = StructVector::GetEntries(ListVector::GetEntry(mid));
    k1 = FlatVector::GetData<int32_t>(*fields[0]);
    if (two_keys) k2 = FlatVector::GetData<int32_t>(*fields[1]);
    val = FlatVector::GetData<float>(*fields[two_keys ? 2 : 1]);
  };
  int32_t *unused = nullptr;
  lists(*children[3], false, tv.lc_outer, tv.lc_sub, tv.lc_key, unused, tv.lc_val, tv.lc_subs, tv.lc_cap);
  if (!nb) {
    lists(*children[4], false, tv.nc_outer, tv.nc_sub, tv.nc_key, unused, tv.nc_val, tv.nc_subs, tv.nc_cap);
    lists(*children[5], true, tv.cc_outer, tv.cc_sub, tv.cc_key1, tv.cc_key2, tv.cc_val, tv.cc_subs, tv.cc_cap);
  }
  if (count == 0) return;
  const uint64_t n = tv.lin_e[1], m = tv.lc_outer[1];
  tv.n = (int32_t)n; tv.m = (int32_t)m;
  for (idx_t i = 0; i < count; i++) {
    bool ok = tv.lin_e[2 * i + 1] == n && tv.lc_outer[2 * i + 1] == m && tv.quad_e[2 * i + 1] == (nb ? n : n * (n + 1) / 2);
    if (!nb) ok = ok && tv.nc_outer[2 * i + 1] == n * m && tv.cc_outer[2 * i + 1] == m * (m + 1) / 2;
    if (!ok) throw InvalidInputException("triple argument: list lengths do not match its lin_agg / lin_cat lengths");
  }
}

// The result STRUCT vector sized for `rows` triples of shape (n, m) whose payload lists hold
// lc / nc / cc entries, as a cofactor_tvec the kernels' results are copied into.
static void ResultAsTvec(Vector &result, idx_t rows, idx_t n, idx_t m, bool nb, idx_t lc, idx_t nc, idx_t cc, cofactor_tvec &tv) {
  auto &children = StructVector::GetEntries(result);
  const idx_t T = nb ? n : n * (n + 1) / 2;
  ReserveList(*children[1], rows * n);
  ReserveList(*children[2], rows * T);
  auto size_lists = [&](Vector &outer, idx_t subs, idx_t payload) {
    ReserveList(outer, subs);
    ReserveList(ListVector::GetEntry(outer), payload);
  };
  size_lists(*children[3], rows * m, lc);
  if (!nb) { size_lists(*children[4], rows * n * m, nc); size_lists(*children[5], rows * m * (m + 1) / 2, cc); }
  VectorAsTvec(result, 0, nb, tv);                    // pointers only (fetched after the Reserves)
  tv.count = rows; tv.n = (int32_t)n; tv.m = (int32_t)m;
  tv.lc_cap = lc; tv.nc_cap = nc; tv.cc_cap = cc;
}

// update of sum_triple / sum_nb_agg: Triple::Sum (sum.cpp:57-261), sum_nb_agg (sum_nb_agg.cpp:45-175)
template <bool NB>
static void RingUpdateTriples(Vector inputs[], AggregateInputData &, idx_t input_count, Vector &state_vector, idx_t count) {
  D_ASSERT(input_count == 1);
  UnifiedVectorFormat sdata;
  state_vector.ToUnifiedFormat(count, sdata);
  auto states = (RingState **)sdata.data;
  RecursiveFlatten(inputs[0], count);
  RingState *first = states[sdata.sel->get_index(0)];
  bool single = true;
  for (idx_t i = 1; i < count && single; i++) single = states[sdata.sel->get_index(i)] == first;
  if (single) {                                       // the whole chunk into one state: one device call
    cofactor_tvec tv;
    VectorAsTvec(inputs[0], count, NB, tv);
    if (!first->agg) Check(cofactor_agg_create(Context(), tv.n, tv.m, NB ? COFACTOR_NB : COFACTOR_TRIPLE, &first->agg));
    Check(cofactor_agg_update_tvec_host(first->agg, &tv));
    return;
  }
  std::vector<double> blob;                           // GROUP BY over triples: row by row
  for (idx_t i = 0; i < count; i++) {
    blob.clear();
    RowToBlob(inputs[0], i, NB, blob);
    RingState *st = states[sdata.sel->get_index(i)];
    if (!st->agg)
      Check(cofactor_agg_create(Context(), (int)blob[1], (int)blob[2], NB ? COFACTOR_NB : COFACTOR_TRIPLE, &st->agg));
    const uint64_t offs[2] = {0, blob.size()};
    Check(cofactor_agg_update_triples(st->agg, blob.data(), offs, 1));
  }
}

// combine: Triple::SumStateCombine (sum_state.cpp:10-114)
// one group's triple as a flat blob: what its aggregate and / or its pool slot hold
static bool StateBlob(RingState *st, std::vector<double> &blob) {
  std::vector<double> a, b;
  uint64_t need = 0;
  if (st->agg) {
    Check(cofactor_agg_finalize(st->agg, nullptr, 0, &need));
    a.resize(need);
    Check(cofactor_agg_finalize(st->agg, a.data(), need, &need));
  }
  if (st->slot >= 0) {
    std::lock_guard<std::mutex> lock(st->pool->mu);
    Check(cofactor_groups_finalize(st->pool->grp, st->slot, nullptr, 0, &need));
    b.resize(need);
    Check(cofactor_groups_finalize(st->pool->grp, st->slot, b.data(), need, &need));
  }
  if (a.empty() && b.empty()) return false;
  if (a.empty() || b.empty()) { blob = a.empty() ? std::move(b) : std::move(a); return true; }
  Check(cofactor_triple_add(a.data(), a.size(), b.data(), b.size(), nullptr, 0, &need));
  blob.resize(need);
  Check(cofactor_triple_add(a.data(), a.size(), b.data(), b.size(), blob.data(), need, &need));
  return true;
}

static void RingCombine(Vector &state, Vector &combined, AggregateInputData &, idx_t count) {
  UnifiedVectorFormat sdata;
  state.ToUnifiedFormat(count, sdata);
  auto src = (RingState **)sdata.data;
  auto dst = FlatVector::GetData<RingState *>(combined);
  for (idx_t i = 0; i < count; i++) {
    RingState *s = src[sdata.sel->get_index(i)], *d = dst[i];
    if (s->slot >= 0) {                               // pool slots: one row add on the device
      if (d->slot < 0) { d->slot = s->slot; d->pool = s->pool; s->slot = -1; }
      else if (d->pool == s->pool) {
        std::lock_guard<std::mutex> lock(d->pool->mu);
        Check(cofactor_groups_combine(d->pool->grp, d->slot, s->slot));
      } else {                                        // slots of different pools: through the blob
        std::vector<double> blob;
        RingState only_slot = {nullptr, s->pool, s->slot};
        if (StateBlob(&only_slot, blob)) {
          if (!d->agg) Check(cofactor_agg_create(Context(), (int)blob[1], (int)blob[2], blob[0] != 0 ? COFACTOR_NB : COFACTOR_TRIPLE, &d->agg));
          const uint64_t offs[2] = {0, blob.size()};
          Check(cofactor_agg_update_triples(d->agg, blob.data(), offs, 1));
        }
      }
    }
    if (!s->agg) continue;                            // nothing (more) in the thread-local state
    if (!d->agg) { d->agg = s->agg; s->agg = nullptr; continue; }   // adopt
    Check(cofactor_agg_combine(d->agg, s->agg));
  }
}

// finalize: Triple::SumStateFinalize (sum_state.cpp:116-464)
static void RingFinalize(Vector &state_vector, AggregateInputData &, Vector &result, idx_t count, idx_t offset) {
  D_ASSERT(offset == 0);                              // as the reference asserts (:120)
  UnifiedVectorFormat sdata;
  state_vector.ToUnifiedFormat(count, sdata);
  auto states = (RingState **)sdata.data;
  ListCursor cur;
  std::vector<double> blob;
  for (idx_t i = 0; i < count; i++) {
    RingState *st = states[sdata.sel->get_index(i)];
    if (!StateBlob(st, blob)) { FlatVector::SetNull(result, i + offset, true); continue; }
    BlobToRow(blob.data(), result, i + offset, cur);
  }
}

template <bool NB>
static unique_ptr<FunctionData> RingAggregateBind(ClientContext &, AggregateFunction &function,
                                                  vector<unique_ptr<Expression>> &) {
  function.return_type = TripleType(NB, /*aggregate_names=*/true);
  return make_uniq<RingBindData>(function.return_type);
}

// ---- scalar callbacks -----------------------------------------------------------------------------
// to_cofactor / to_nb_agg: Triple::CustomLift (lift.cpp:15-243), to_nb_lift (lift_to_nb_agg.cpp:13-136)
template <bool NB>
static void LiftFunction(DataChunk &args, ExpressionState &, Vector &result) {
  const idx_t rows = args.size();
  vector<const float *> num;
  vector<const int32_t *> cat;
  for (idx_t j = 0; j < args.ColumnCount(); j++) {
    args.data[j].Flatten(rows);
    if (IsNumeric(args.data[j].GetType())) num.push_back(FlatVector::GetData<float>(args.data[j]));
    else cat.push_back(FlatVector::GetData<int32_t>(args.data[j]));
  }
  // the expand kernel writes the result's child arrays (regular shape: one entry per sub-list)
  const idx_t n = num.size(), m = cat.size();
  result.SetVectorType(VectorType::FLAT_VECTOR);
  cofactor_tvec out;
  ResultAsTvec(result, rows, n, m, NB, rows * m, NB ? 0 : rows * n * m, NB ? 0 : rows * m * (m + 1) / 2, out);
  Check(cofactor_lift_host_tvec(Context(), num.data(), (int)n, cat.data(), (int)m, rows, NB ? COFACTOR_NB : COFACTOR_TRIPLE, &out));
}

// multiply_triple / multiply_nb_agg: Triple::MultiplyFunction (mul.cpp:19-611), multiply_nb (mul_nb.cpp:20-268)
template <bool NB>
static void MultiplyFunction(DataChunk &args, ExpressionState &, Vector &result) {
  const idx_t rows = args.size();
  RecursiveFlatten(args.data[0], rows);
  RecursiveFlatten(args.data[1], rows);
  result.SetVectorType(VectorType::FLAT_VECTOR);
  cofactor_tvec a, b, out;
  VectorAsTvec(args.data[0], rows, NB, a);
  VectorAsTvec(args.data[1], rows, NB, b);
  uint64_t lc = 0, nc = 0, cc = 0;                    // two-call protocol: payload sizes first
  Check(cofactor_multiply_host(Context(), &a, nullptr, &b, nullptr, rows, nullptr, &lc, &nc, &cc));
  ResultAsTvec(result, rows, a.n + b.n, a.m + b.m, NB, lc, nc, cc, out);
  Check(cofactor_multiply_host(Context(), &a, nullptr, &b, nullptr, rows, &out, nullptr, nullptr, nullptr));
}

template <bool NB>
static unique_ptr<FunctionData> RingScalarBind(ClientContext &, ScalarFunction &function,
                                               vector<unique_ptr<Expression>> &) {
  function.return_type = TripleType(NB, /*aggregate_names=*/false);
  return make_uniq<VariableReturnBindData>(function.return_type);
}

// ---- registration (reference: duckdb_imputation_extension.cpp:48-180) ------------------------------
template <bool NB>
static void LoadRing(DatabaseInstance &instance) {
  using SF = AggregateFunction;
  auto sum_lifted = SF(NB ? "sum_nb_agg" : "sum_triple", {LogicalType::ANY}, LogicalTypeId::STRUCT,
                       SF::StateSize<RingState>, SF::StateInitialize<RingState, RingStateFunction>,
                       RingUpdateTriples<NB>, RingCombine, RingFinalize, nullptr, RingAggregateBind<NB>,
                       SF::StateDestroy<RingState, RingStateFunction>, nullptr, nullptr);
  ExtensionUtil::RegisterFunction(instance, sum_lifted);

  ScalarFunction lift(NB ? "to_nb_agg" : "to_cofactor", {}, LogicalTypeId::STRUCT, LiftFunction<NB>, RingScalarBind<NB>);
  lift.varargs = LogicalType::ANY;
  lift.null_handling = FunctionNullHandling::SPECIAL_HANDLING;
  lift.serialize = VariableReturnBindData::Serialize;
  lift.deserialize = VariableReturnBindData::Deserialize;
  ExtensionUtil::RegisterFunction(instance, lift);

  ScalarFunction mul(NB ? "multiply_nb_agg" : "multiply_triple", {LogicalType::ANY}, LogicalTypeId::STRUCT,
                     MultiplyFunction<NB>, RingScalarBind<NB>);
  mul.varargs = LogicalType::ANY;
  mul.null_handling = FunctionNullHandling::SPECIAL_HANDLING;
  mul.serialize = VariableReturnBindData::Serialize;
  mul.deserialize = VariableReturnBindData::Deserialize;
  ExtensionUtil::RegisterFunction(instance, mul);

  // 0..20 inclusive (the reference stops at 19; BASELINE.json's metric is sum_to_triple_20_0)
  for (int x = 0; x <= COFACTOR_MAX_NUM; x++)
    for (int y = 0; y <= COFACTOR_MAX_CAT; y++) {
      if (x == 0 && y == 0) continue;
      vector<LogicalType> argt;
      for (int i = 0; i < x; i++) argt.push_back(LogicalType::FLOAT);
      for (int i = 0; i < y; i++) argt.push_back(LogicalType::INTEGER);
      auto fn = SF(std::string(NB ? "sum_to_nb_agg_" : "sum_to_triple_") + std::to_string(x) + "_" + std::to_string(y),
                   argt, LogicalTypeId::STRUCT, SF::StateSize<RingState>,
                   SF::StateInitialize<RingState, RingStateFunction>, RingUpdate<NB>, RingCombine, RingFinalize,
                   nullptr, RingAggregateBind<NB>, SF::StateDestroy<RingState, RingStateFunction>, nullptr, nullptr);
      fn.varargs = LogicalType::ANY;
      fn.null_handling = FunctionNullHandling::SPECIAL_HANDLING;
      ExtensionUtil::RegisterFunction(instance, fn);
    }
}

// ---- consumers of the triple (reference: load_ml, duckdb_imputation_extension.cpp:182-249) ---------
static unique_ptr<FunctionData> FloatListBind(ClientContext &, ScalarFunction &function,
                                              vector<unique_ptr<Expression>> &) {
  function.return_type = LogicalType::LIST(LogicalType::FLOAT);      // regression.cpp:356-362, lda.cpp:154-159
  return make_uniq<VariableReturnBindData>(function.return_type);
}

template <class T>
static T ConstArg(DataChunk &args, idx_t col, const char *fn) {     // args.data[i].GetValue(0).GetValue<T>()
  if (col >= args.ColumnCount()) throw InvalidInputException("%s: too few arguments", fn);
  return args.data[col].GetValue(0).GetValue<T>();
}

// one parameter vector as a constant LIST(FLOAT) (regression.cpp:288-312)
static void EmitFloatList(Vector &result, const std::vector<float> &v) {
  result.SetVectorType(VectorType::CONSTANT_VECTOR);
  ListVector::Reserve(result, v.size());
  ListVector::SetListSize(result, v.size());
  auto out = FlatVector::GetData<float>(ListVector::GetEntry(result));
  for (idx_t i = 0; i < v.size(); i++) out[i] = v[i];
  auto meta = ListVector::GetData(result);
  meta[0].offset = 0;
  meta[0].length = v.size();
}

// linreg_train(triple, label, step_size, lambda, max_iterations, compute_variance, normalize)
static void LinregTrain(DataChunk &args, ExpressionState &, Vector &result) {
  RecursiveFlatten(args.data[0], args.size());
  std::vector<double> blob;
  RowToBlob(args.data[0], 0, false, blob);
  const int label = ConstArg<int32_t>(args, 1, "linreg_train");
  const float step = ConstArg<float>(args, 2, "linreg_train"), lambda = ConstArg<float>(args, 3, "linreg_train");
  const int iters = ConstArg<int32_t>(args, 4, "linreg_train");
  const bool variance = ConstArg<bool>(args, 5, "linreg_train"), normalize = ConstArg<bool>(args, 6, "linreg_train");
  uint64_t need = 0;
  Check(cofactor_linreg_train(blob.data(), blob.size(), label, step, lambda, iters, variance, normalize, nullptr, 0, &need));
  std::vector<float> params(need);
  Check(cofactor_linreg_train(blob.data(), blob.size(), label, step, lambda, iters, variance, normalize, params.data(), need, &need));
  EmitFloatList(result, params);
}

// lda_train(triple, label, shrinkage, normalize)
static void LdaTrain(DataChunk &args, ExpressionState &, Vector &result) {
  RecursiveFlatten(args.data[0], args.size());
  std::vector<double> blob;
  RowToBlob(args.data[0], 0, false, blob);
  const int label = ConstArg<int32_t>(args, 1, "lda_train");
  const float shrinkage = ConstArg<float>(args, 2, "lda_train");
  const bool normalize = ConstArg<bool>(args, 3, "lda_train");
  uint64_t need = 0;
  Check(cofactor_lda_train(blob.data(), blob.size(), label, shrinkage, normalize, nullptr, 0, &need));
  std::vector<float> params(need);
  Check(cofactor_lda_train(blob.data(), blob.size(), label, shrinkage, normalize, params.data(), need, &need));
  EmitFloatList(result, params);
}

// the FLOAT[] parameter argument (a constant list) and the feature columns after `first`
struct PredictArgs {
  std::vector<float> params;
  vector<const float *> num;
  vector<const int32_t *> cat;
  PredictArgs(DataChunk &args, idx_t first) {
    const idx_t rows = args.size();
    RecursiveFlatten(args.data[0], rows);
    auto meta = ListVector::GetData(args.data[0]);
    auto child = FlatVector::GetData<float>(ListVector::GetEntry(args.data[0]));
    params.assign(child + meta[0].offset, child + meta[0].offset + meta[0].length);
    for (idx_t j = first; j < args.ColumnCount(); j++) {
      args.data[j].Flatten(rows);
      if (args.data[j].GetType() == LogicalType::FLOAT) num.push_back(FlatVector::GetData<float>(args.data[j]));
      else if (args.data[j].GetType() == LogicalType::INTEGER) cat.push_back(FlatVector::GetData<int32_t>(args.data[j]));
      else throw InvalidInputException("predict: feature columns must be FLOAT or INTEGER");
    }
  }
};

static uint64_t NoiseSeed() {     // the reference seeds random() off /dev/urandom once (regression.cpp:377-395)
  static std::atomic<uint64_t> next{std::random_device{}() * 0x9E3779B97F4A7C15ull};
  return next.fetch_add(0x632BE59BD9B4E019ull);
}

// linreg_predict(params, noise, normalize, feature columns...)
static void LinregPredict(DataChunk &args, ExpressionState &, Vector &result) {
  const bool noise = ConstArg<bool>(args, 1, "linreg_predict"), normalize = ConstArg<bool>(args, 2, "linreg_predict");
  PredictArgs in(args, 3);
  result.SetVectorType(VectorType::FLAT_VECTOR);
  Check(cofactor_linreg_predict_host(Context(), in.params.data(), in.params.size(), noise, normalize,
                                     noise ? NoiseSeed() : 0, in.num.data(), (int32_t)in.num.size(), in.cat.data(),
                                     (int32_t)in.cat.size(), args.size(), FlatVector::GetData<float>(result)));
}

// lda_predict(params, normalize, feature columns...) -> class index (lda.cpp:560)
static void LdaPredict(DataChunk &args, ExpressionState &, Vector &result) {
  const bool normalize = ConstArg<bool>(args, 1, "lda_predict");
  PredictArgs in(args, 2);
  result.SetVectorType(VectorType::FLAT_VECTOR);
  Check(cofactor_lda_predict_host(Context(), in.params.data(), in.params.size(), normalize, /*emit_label=*/0,
                                  in.num.data(), (int32_t)in.num.size(), in.cat.data(), (int32_t)in.cat.size(),
                                  args.size(), FlatVector::GetData<int32_t>(result)));
}

static unique_ptr<FunctionData> FloatBind(ClientContext &, ScalarFunction &function, vector<unique_ptr<Expression>> &) {
  function.return_type = LogicalType::FLOAT;                          // regression.cpp:365-374
  function.varargs = LogicalType::ANY;
  return make_uniq<VariableReturnBindData>(function.return_type);
}
static unique_ptr<FunctionData> IntegerBind(ClientContext &, ScalarFunction &function, vector<unique_ptr<Expression>> &) {
  function.return_type = LogicalType::INTEGER;                        // lda.cpp:593-601
  function.varargs = LogicalType::ANY;
  return make_uniq<VariableReturnBindData>(function.return_type);
}

static void LoadML(DatabaseInstance &instance) {
  auto reg = [&](const char *name, LogicalTypeId ret, scalar_function_t fn, bind_scalar_function_t bind) {
    ScalarFunction f(name, {LogicalType::ANY}, ret, fn, bind);
    f.varargs = LogicalType::ANY;
    f.null_handling = FunctionNullHandling::SPECIAL_HANDLING;
    f.serialize = VariableReturnBindData::Serialize;
    f.deserialize = VariableReturnBindData::Deserialize;
    ExtensionUtil::RegisterFunction(instance, f);
  };
  reg("linreg_train", LogicalTypeId::LIST, LinregTrain, FloatListBind);
  reg("lda_train", LogicalTypeId::LIST, LdaTrain, FloatListBind);
  reg("linreg_predict", LogicalTypeId::INTEGER, LinregPredict, FloatBind);
  reg("lda_predict", LogicalTypeId::INTEGER, LdaPredict, IntegerBind);
}

}  // namespace cofactor_glue

void DuckdbImputationExtension::Load(DuckDB &db) {
  cofactor_glue::LoadRing<false>(*db.instance);
  cofactor_glue::LoadRing<true>(*db.instance);
  cofactor_glue::LoadML(*db.instance);
  // qda_* / nb_* consume finalised triples too and are outside this library's scope (DESIGN.md
  // §8); a build that wants them links the reference's ML/qda.cpp and ML/naive_bayes.cpp.
}
std::string DuckdbImputationExtension::Name() { return "duckdb_imputation"; }

}  // namespace duckdb

extern "C" {
DUCKDB_EXTENSION_API void duckdb_imputation_init(duckdb::DatabaseInstance &db) {
  duckdb::DuckDB db_wrapper(db);
  db_wrapper.LoadExtension<duckdb::DuckdbImputationExtension>();
}
DUCKDB_EXTENSION_API const char *duckdb_imputation_version() { return duckdb::DuckDB::LibraryVersion(); }
}

#ifndef DUCKDB_EXTENSION_MAIN
#error DUCKDB_EXTENSION_MAIN not defined
#endif
