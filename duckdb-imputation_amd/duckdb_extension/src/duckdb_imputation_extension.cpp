// DuckDB (v0.9.2 C++ extension API) glue for libcofactor_hip: registers the same SQL functions,
// argument types and STRUCT return types as the reference's ring code
// (reference: duckdb_extension/src/duckdb_imputation_extension.cpp:48-180) and forwards every
// callback across the C ABI of include/cofactor_hip.h.
//
//   sum_to_triple_<x>_<y>(FLOAT*x, INTEGER*y)   x,y in 0..20      -> update_host / combine / finalize
//   sum_to_nb_agg_<x>_<y>(...)                                     -> same, kind = NB
//   sum_triple(triple), sum_nb_agg(triple)                          -> update_triples
//   to_cofactor(cols...), to_nb_agg(cols...)                        -> cofactor_lift_host
//   multiply_triple(a, b), multiply_nb_agg(a, b)                    -> cofactor_triple_multiply
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

// ---- one GPU context per process ----------------------------------------------------------------
static cofactor_ctx *Context() {
  static std::once_flag once;
  static cofactor_ctx *ctx = nullptr;
  std::call_once(once, [] {
    const char *dev = std::getenv("COFACTOR_DEVICE");
    if (cofactor_ctx_create(dev ? std::atoi(dev) : 0, &ctx) != COFACTOR_OK)
      throw IOException("duckdb_imputation (MI355X): %s", cofactor_last_error());
  });
  return ctx;
}

static void Check(cofactor_status st) {
  if (st != COFACTOR_OK) throw InvalidInputException("duckdb_imputation (MI355X): %s", cofactor_last_error());
}

// ---- aggregate state: a POD handle, memcpy-relocatable like Triple::SumState ---------------------
// (reference: duckdb_extension/src/include/triple/sum/sum_state.h:14-57)
struct RingState {
  cofactor_agg *agg;
};

struct RingStateFunction {
  template <class STATE>
  static void Initialize(STATE &state) { state.agg = nullptr; }
  template <class STATE>
  static void Destroy(STATE &state, AggregateInputData &) {
    cofactor_agg_destroy(state.agg);
    state.agg = nullptr;
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
static void RingUpdate(Vector inputs[], AggregateInputData &, idx_t cols, Vector &state_vector, idx_t count) {
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
  // rows of this chunk per distinct state (GROUP BY); one state in the ungrouped case
  std::unordered_map<RingState *, vector<uint32_t>> rows_of;
  RingState *first = states[sdata.sel->get_index(0)];
  bool single = true;
  for (idx_t i = 1; i < count && single; i++) single = states[sdata.sel->get_index(i)] == first;
  auto feed = [&](RingState *st, const uint32_t *row_idx, idx_t rows) {
    if (!st->agg)
      Check(cofactor_agg_create(Context(), (int)num.size(), (int)cat.size(), NB ? COFACTOR_NB : COFACTOR_TRIPLE, &st->agg));
    Check(cofactor_agg_update_host(st->agg, num.data(), cat.data(), num_sel.data(), cat_sel.data(), row_idx, rows));
  };
  if (single) { feed(first, nullptr, count); return; }
  for (idx_t i = 0; i < count; i++) rows_of[states[sdata.sel->get_index(i)]].push_back((uint32_t)i);
  for (auto &kv : rows_of) feed(kv.first, kv.second.data(), kv.second.size());
}

// update of sum_triple / sum_nb_agg: Triple::Sum (sum.cpp:57-261), sum_nb_agg (sum_nb_agg.cpp:45-175)
template <bool NB>
static void RingUpdateTriples(Vector inputs[], AggregateInputData &, idx_t input_count, Vector &state_vector, idx_t count) {
  D_ASSERT(input_count == 1);
  UnifiedVectorFormat sdata;
  state_vector.ToUnifiedFormat(count, sdata);
  auto states = (RingState **)sdata.data;
  RecursiveFlatten(inputs[0], count);
  std::vector<double> blob;
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
static void RingCombine(Vector &state, Vector &combined, AggregateInputData &, idx_t count) {
  UnifiedVectorFormat sdata;
  state.ToUnifiedFormat(count, sdata);
  auto src = (RingState **)sdata.data;
  auto dst = FlatVector::GetData<RingState *>(combined);
  for (idx_t i = 0; i < count; i++) {
    RingState *s = src[sdata.sel->get_index(i)];
    if (!s->agg) continue;                            // empty thread-local state
    if (!dst[i]->agg) { dst[i]->agg = s->agg; s->agg = nullptr; continue; }   // adopt
    Check(cofactor_agg_combine(dst[i]->agg, s->agg));
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
    if (!st->agg) { FlatVector::SetNull(result, i + offset, true); continue; }
    uint64_t need = 0;
    Check(cofactor_agg_finalize(st->agg, nullptr, 0, &need));
    blob.resize(need);
    Check(cofactor_agg_finalize(st->agg, blob.data(), need, &need));
    BlobToRow(blob.data(), result, i + offset, cur);
  }
}

template <bool NB>
static unique_ptr<FunctionData> RingAggregateBind(ClientContext &, AggregateFunction &function,
                                                  vector<unique_ptr<Expression>> &) {
  function.return_type = TripleType(NB, /*aggregate_names=*/true);
  return make_uniq<VariableReturnBindData>(function.return_type);
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
  uint64_t need = 0;
  std::vector<uint64_t> offs(rows + 1);
  Check(cofactor_lift_host(num.data(), (int)num.size(), cat.data(), (int)cat.size(), rows,
                           NB ? COFACTOR_NB : COFACTOR_TRIPLE, nullptr, 0, &need, nullptr));
  std::vector<double> blobs(need);
  Check(cofactor_lift_host(num.data(), (int)num.size(), cat.data(), (int)cat.size(), rows,
                           NB ? COFACTOR_NB : COFACTOR_TRIPLE, blobs.data(), need, &need, offs.data()));
  ListCursor cur;
  result.SetVectorType(VectorType::FLAT_VECTOR);
  for (idx_t i = 0; i < rows; i++) BlobToRow(blobs.data() + offs[i], result, i, cur);
}

// multiply_triple / multiply_nb_agg: Triple::MultiplyFunction (mul.cpp:19-611), multiply_nb (mul_nb.cpp:20-268)
template <bool NB>
static void MultiplyFunction(DataChunk &args, ExpressionState &, Vector &result) {
  const idx_t rows = args.size();
  RecursiveFlatten(args.data[0], rows);
  RecursiveFlatten(args.data[1], rows);
  ListCursor cur;
  std::vector<double> a, b, out;
  result.SetVectorType(VectorType::FLAT_VECTOR);
  for (idx_t i = 0; i < rows; i++) {
    a.clear(); b.clear();
    RowToBlob(args.data[0], i, NB, a);
    RowToBlob(args.data[1], i, NB, b);
    uint64_t need = 0;
    Check(cofactor_triple_multiply(a.data(), a.size(), b.data(), b.size(), nullptr, 0, &need));
    out.resize(need);
    Check(cofactor_triple_multiply(a.data(), a.size(), b.data(), b.size(), out.data(), need, &need));
    BlobToRow(out.data(), result, i, cur);
  }
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
