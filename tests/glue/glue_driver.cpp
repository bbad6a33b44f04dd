// Drives OUR DuckDB glue (duckdb-imputation_amd/duckdb_extension/src/duckdb_imputation_extension.cpp)
// through the callback sequence DuckDB's executor uses, against the test stand-in of the DuckDB
// API in duckdb_stub/ (the image has no DuckDB).  Prints one JSON document; tests/test_glue.py
// compares it with the reference's golden literals.  Needs a GPU (update goes to the HIP kernels).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <iomanip>
#include <sstream>
#include <thread>
#include <string>

#include "duckdb.hpp"

extern "C" void duckdb_imputation_init(duckdb::DatabaseInstance &db);
extern "C" const char *duckdb_imputation_version();

using namespace duckdb;

// ---- the 5-row table of the reference's tests (test_sum.py:15-16) -------------------------------
static const int GB[5] = {1, 1, 2, 2, 2};
static const float A_[5] = {1, 5, 2, 5, 2}, B_[5] = {2, 6, 1, 7, 1}, C_[5] = {3, 7, 3, 6, 3};
static const int D_[5] = {4, 8, 4, 8, 4}, E_[5] = {5, 9, 6, 10, 6}, F_[5] = {6, 10, 8, 12, 8};

static Vector FloatCol(const float *v, const std::vector<int> &rows, bool dictionary) {
  Vector out(LogicalType(LogicalType::FLOAT), 64);
  auto d = FlatVector::GetData<float>(out);
  if (!dictionary) {
    for (size_t i = 0; i < rows.size(); i++) d[i] = v[rows[i]];
  } else {  // physical buffer holds the whole column reversed; the selection picks the wanted rows
    for (int i = 0; i < 5; i++) d[4 - i] = v[i];
    auto sel = std::make_shared<std::vector<sel_t>>();
    for (int r : rows) sel->push_back((sel_t)(4 - r));
    out.Slice(sel);
  }
  return out;
}
static Vector IntCol(const int *v, const std::vector<int> &rows, bool dictionary) {
  Vector out(LogicalType(LogicalType::INTEGER), 64);
  auto d = FlatVector::GetData<int32_t>(out);
  if (!dictionary) {
    for (size_t i = 0; i < rows.size(); i++) d[i] = v[rows[i]];
  } else {
    for (int i = 0; i < 5; i++) d[4 - i] = v[i];
    auto sel = std::make_shared<std::vector<sel_t>>();
    for (int r : rows) sel->push_back((sel_t)(4 - r));
    out.Slice(sel);
  }
  return out;
}

// ---- JSON dump of row `row` of a triple STRUCT vector, field names from its type ----------------
static void DumpValue(Vector &v, idx_t row, std::ostringstream &o);
static void DumpList(Vector &v, idx_t row, std::ostringstream &o) {
  auto e = ListVector::GetData(v)[row];
  Vector &child = ListVector::GetEntry(v);
  o << "[";
  for (idx_t i = 0; i < e.length; i++) { if (i) o << ","; DumpValue(child, e.offset + i, o); }
  o << "]";
}
static void DumpValue(Vector &v, idx_t row, std::ostringstream &o) {
  switch (v.GetType().id()) {
  case LogicalTypeId::INTEGER: o << FlatVector::GetData<int32_t>(v)[row]; break;
  case LogicalTypeId::FLOAT: { char b[64]; snprintf(b, sizeof(b), "%.9g", (double)FlatVector::GetData<float>(v)[row]); o << b; break; }
  case LogicalTypeId::LIST: DumpList(v, row, o); break;
  case LogicalTypeId::STRUCT: {
    auto &fields = StructType::GetChildTypes(v.GetType());
    auto &entries = StructVector::GetEntries(v);
    o << "{";
    for (size_t f = 0; f < fields.size(); f++) {
      if (f) o << ",";
      o << "\"" << fields[f].first << "\":";
      DumpValue(*entries[f], row, o);
    }
    o << "}";
    break;
  }
  default: o << "null";
  }
}

// ---- a miniature hash-aggregate: groups -> states, update per chunk, combine, finalize ----------
struct AggRun {
  AggregateFunction &fn;
  LogicalType result_type;
  std::vector<std::vector<data_t>> state_mem;   // one state per group
  AggregateInputData aid;
  ClientContext ctx;
  std::shared_ptr<FunctionData> bind_data;      // one bind per query, shared by its threads' runs
  AggRun(AggregateFunction &f, idx_t groups, std::shared_ptr<FunctionData> shared = nullptr) : fn(f) {
    vector<unique_ptr<Expression>> args;
    if (shared) bind_data = shared;
    else bind_data = std::shared_ptr<FunctionData>(fn.bind(ctx, fn, args).release());
    aid.bind_data = bind_data.get();
    result_type = fn.return_type;
    state_mem.assign(groups, std::vector<data_t>(fn.state_size(), 0xAB));
    for (auto &m : state_mem) fn.initialize(m.data());
  }
  // rows_group[i] = group of logical row i of this chunk
  void Update(std::vector<Vector> &cols, const std::vector<int> &rows_group) {
    Vector states(LogicalType(LogicalType::ANY), 64);
    auto sp = FlatVector::GetData<data_ptr_t>(states);
    for (size_t i = 0; i < rows_group.size(); i++) sp[i] = state_mem[rows_group[i]].data();
    fn.update(cols.data(), aid, cols.size(), states, rows_group.size());
  }
  void CombineInto(AggRun &target) {            // thread-local states -> global states
    Vector src(LogicalType(LogicalType::ANY), 64), dst(LogicalType(LogicalType::ANY), 64);
    for (size_t g = 0; g < state_mem.size(); g++) {
      FlatVector::GetData<data_ptr_t>(src)[g] = state_mem[g].data();
      FlatVector::GetData<data_ptr_t>(dst)[g] = target.state_mem[g].data();
    }
    fn.combine(src, dst, aid, state_mem.size());
  }
  Vector Finalize() {
    Vector states(LogicalType(LogicalType::ANY), 64);
    for (size_t g = 0; g < state_mem.size(); g++) FlatVector::GetData<data_ptr_t>(states)[g] = state_mem[g].data();
    Vector result(result_type, 64);
    fn.finalize(states, aid, result, state_mem.size(), 0);
    return result;
  }
  void DestroyStates() {                        // what DuckDB does with a hash table's states when the query ends
    Vector states(LogicalType(LogicalType::ANY), 64);
    for (size_t g = 0; g < state_mem.size(); g++) FlatVector::GetData<data_ptr_t>(states)[g] = state_mem[g].data();
    if (fn.destructor) fn.destructor(states, aid, state_mem.size());
    state_mem.clear();
  }
  ~AggRun() { DestroyStates(); }
};

static std::vector<Vector> Cols(const std::string &names, const std::vector<int> &rows, bool dict) {
  std::vector<Vector> out;
  for (char c : names) {
    switch (c) {
    case 'a': out.push_back(FloatCol(A_, rows, dict)); break;
    case 'b': out.push_back(FloatCol(B_, rows, dict)); break;
    case 'c': out.push_back(FloatCol(C_, rows, dict)); break;
    case 'd': out.push_back(IntCol(D_, rows, dict)); break;
    case 'e': out.push_back(IntCol(E_, rows, dict)); break;
    case 'f': out.push_back(IntCol(F_, rows, dict)); break;
    }
  }
  return out;
}

static std::string Rows(Vector &result, idx_t count) {
  std::ostringstream o;
  o << "[";
  for (idx_t i = 0; i < count; i++) { if (i) o << ","; DumpValue(result, i, o); }
  o << "]";
  return o.str();
}

int main() {
  DatabaseInstance db;
  try {
    duckdb_imputation_init(db);
    std::ostringstream out;
    out << "{\"version\":\"" << duckdb_imputation_version() << "\"";
    out << ",\"n_aggregates\":" << db.aggregates.size() << ",\"n_scalars\":" << db.scalars.size();
    out << ",\"has\":{";
    const char *names[] = {"sum_triple", "sum_nb_agg", "sum_to_triple_20_0", "sum_to_triple_0_20", "sum_to_triple_20_20",
                           "sum_to_nb_agg_20_20", "sum_to_triple_0_0"};
    for (size_t i = 0; i < sizeof(names) / sizeof(*names); i++)
      out << (i ? "," : "") << "\"" << names[i] << "\":" << (db.aggregates.count(names[i]) ? "true" : "false");
    const char *snames[] = {"to_cofactor", "to_nb_agg", "multiply_triple", "multiply_nb_agg",
                            "linreg_train", "linreg_predict", "lda_train", "lda_predict"};
    for (auto n : snames) out << ",\"" << n << "\":" << (db.scalars.count(n) ? "true" : "false");
    out << "}";

    const std::vector<int> all = {0, 1, 2, 3, 4};
    for (int nb = 0; nb < 2; nb++) {
      const std::string pfx = nb ? "nb_" : "";
      auto &fn33 = db.aggregates.at(nb ? "sum_to_nb_agg_3_3" : "sum_to_triple_3_3");
      {  // SELECT sum_to_triple_3_3(a,b,c,d,e,f) FROM test   — one flat chunk, one state
        AggRun run(fn33, 1);
        auto cols = Cols("abcdef", all, false);
        run.Update(cols, {0, 0, 0, 0, 0});
        Vector r = run.Finalize();
        out << ",\"" << pfx << "sum_all\":" << Rows(r, 1);
      }
      {  // ... GROUP BY gb — dictionary vectors, per-row state pointers, two chunks
        AggRun run(fn33, 2);
        auto c1 = Cols("abcdef", {0, 1, 2}, true);
        run.Update(c1, {0, 0, 1});
        auto c2 = Cols("abcdef", {3, 4}, true);
        run.Update(c2, {1, 1});
        Vector r = run.Finalize();
        out << ",\"" << pfx << "sum_group_by\":" << Rows(r, 2);
      }
      {  // a prepared statement executed twice: the same bind data (and GROUP BY pool), fresh states; the
         // states of the first execution are destroyed, their pool slots handed out again — cleared
        AggRun first(fn33, 2);
        std::shared_ptr<FunctionData> bind = first.bind_data;
        auto c1 = Cols("abcdef", {0, 1, 2}, true);
        auto c2 = Cols("abcdef", {3, 4}, true);
        {
          AggRun local(fn33, 2, bind);                 // a thread-local run combined into `first` and destroyed
          local.Update(c1, {0, 0, 1});
          local.CombineInto(first);
        }
        first.Update(c2, {1, 1});
        Vector r1 = first.Finalize();
        std::string once = Rows(r1, 2);
        first.DestroyStates();
        AggRun again(fn33, 2, bind);
        again.Update(c1, {0, 0, 1});
        again.Update(c2, {1, 1});
        Vector r2 = again.Finalize();
        if (Rows(r2, 2) != once) throw std::runtime_error("second execution on the same pool differs from the first");
        out << ",\"" << pfx << "sum_group_by_executed_twice\":" << Rows(r2, 2);
      }
      {  // two worker threads' local states combined into global ones (one of them empty for group 0)
        AggRun global(fn33, 2);
        AggRun t1(fn33, 2, global.bind_data), t2(fn33, 2, global.bind_data);
        auto c1 = Cols("abcdef", {0, 2, 3}, false);
        t1.Update(c1, {0, 1, 1});
        auto c2 = Cols("abcdef", {1, 4}, false);
        t2.Update(c2, {0, 1});
        t1.CombineInto(global);
        t2.CombineInto(global);
        Vector r = global.Finalize();
        out << ",\"" << pfx << "sum_combined\":" << Rows(r, 2);
      }
      {  // the same through a COPY of the bind data (DuckDB copies bound aggregates: optimizer rewrites,
         // CTE inlining) with per-row state pointers, i.e. through the GROUP BY pool the copy must share
        AggRun global(fn33, 2);
        std::shared_ptr<FunctionData> copy(global.bind_data->Copy().release());
        if (!copy->Equals(*global.bind_data)) throw std::runtime_error("a copy of the bind data must equal it");
        AggRun t1(fn33, 2, copy), t2(fn33, 2, global.bind_data);
        auto c1 = Cols("abcdef", {0, 2, 3}, true);
        t1.Update(c1, {0, 1, 1});
        auto c2 = Cols("abcdef", {1, 4}, true);
        t2.Update(c2, {0, 1});
        t1.CombineInto(global);
        t2.CombineInto(global);
        Vector r = global.Finalize();
        out << ",\"" << pfx << "sum_combined_copied_bind\":" << Rows(r, 2);
      }
      {  // two WORKER THREADS: each sticks to its own context (COFACTOR_DEVICES lists two), their
         // thread-local states are merged by combine across the contexts — the extension's multi-GPU seam
        AggRun global(fn33, 1);
        AggRun t1(fn33, 1, global.bind_data), t2(fn33, 1, global.bind_data);
        std::exception_ptr err;
        auto work = [&](AggRun *run, std::vector<int> rows) {
          try {
            auto c = Cols("abcdef", rows, false);
            run->Update(c, std::vector<int>(rows.size(), 0));
          } catch (...) { err = std::current_exception(); }
        };
        std::thread a(work, &t1, std::vector<int>{0, 2, 3}), b(work, &t2, std::vector<int>{1, 4});
        a.join(); b.join();
        if (err) std::rethrow_exception(err);
        t1.CombineInto(global);
        t2.CombineInto(global);
        Vector r = global.Finalize();
        out << ",\"" << pfx << "sum_two_contexts\":" << Rows(r, 1);
      }
      {  // SELECT to_cofactor(a,b,c,d,e,f) FROM test, then sum_triple(...) GROUP BY gb
        auto &lift = db.scalars.at(nb ? "to_nb_agg" : "to_cofactor");
        ClientContext ctx; ExpressionState es;
        vector<unique_ptr<Expression>> args;
        auto bd = lift.bind(ctx, lift, args);
        DataChunk chunk;
        chunk.data = Cols("abcdef", all, true);
        chunk.count = 5;
        Vector lifted(lift.return_type, 64);
        lift.function(chunk, es, lifted);
        out << ",\"" << pfx << "lift_all\":" << Rows(lifted, 5);
        auto &sum = db.aggregates.at(nb ? "sum_nb_agg" : "sum_triple");
        AggRun run(sum, 2);
        std::vector<Vector> in;
        in.push_back(lifted);
        run.Update(in, {0, 0, 1, 1, 1});
        Vector r = run.Finalize();
        out << ",\"" << pfx << "sum_lifted_group_by\":" << Rows(r, 2);
        // ungrouped: the whole chunk of lifted triples into one state (the vector goes to the
        // sum_triple kernels as it is, no per-row blobs)
        AggRun one(sum, 1);
        one.Update(in, {0, 0, 0, 0, 0});
        one.Update(in, {0, 0, 0, 0, 0});
        Vector r1 = one.Finalize();
        out << ",\"" << pfx << "sum_lifted_all_twice\":" << Rows(r1, 1);
      }
      {  // multiply_triple(A, B): A = sum_to_triple_2_2(b,c,d,e) WHERE gb = 1, B = (a,c,d,f) WHERE gb = 2
        auto &fn22 = db.aggregates.at(nb ? "sum_to_nb_agg_2_2" : "sum_to_triple_2_2");
        AggRun ra(fn22, 1), rb(fn22, 1);
        auto ca = Cols("bcde", {0, 1}, false);
        ra.Update(ca, {0, 0});
        auto cb = Cols("acdf", {2, 3, 4}, false);
        rb.Update(cb, {0, 0, 0});
        Vector va = ra.Finalize(), vb = rb.Finalize();
        auto &mul = db.scalars.at(nb ? "multiply_nb_agg" : "multiply_triple");
        ClientContext ctx; ExpressionState es;
        vector<unique_ptr<Expression>> args;
        auto bd = mul.bind(ctx, mul, args);
        DataChunk chunk;
        chunk.data.push_back(va);
        chunk.data.push_back(vb);
        chunk.count = 1;
        Vector prod(mul.return_type, 64);
        mul.function(chunk, es, prod);
        out << ",\"" << pfx << "multiply\":" << Rows(prod, 1);
      }
    }
    {  // the consumers: linreg_train / linreg_predict / lda_train / lda_predict on the 5-row table
      auto constant = [](LogicalTypeId t, double v) {
        Vector c(t, 1);
        c.SetVectorType(VectorType::CONSTANT_VECTOR);
        if (t == LogicalTypeId::BOOLEAN) FlatVector::GetData<uint8_t>(c)[0] = v != 0;
        else if (t == LogicalTypeId::INTEGER) FlatVector::GetData<int32_t>(c)[0] = (int32_t)v;
        else FlatVector::GetData<float>(c)[0] = (float)v;
        return c;
      };
      auto call = [&](const char *name, DataChunk &chunk, idx_t rows) {
        auto &fn = db.scalars.at(name);
        ClientContext ctx; ExpressionState es;
        vector<unique_ptr<Expression>> args;
        auto bd = fn.bind(ctx, fn, args);
        Vector res(fn.return_type, 64);
        chunk.count = rows;
        fn.function(chunk, es, res);
        return res;
      };
      auto floats = [&](Vector &list) {
        std::ostringstream o;
        auto e = ListVector::GetData(list)[0];
        auto v = FlatVector::GetData<float>(ListVector::GetEntry(list));
        o << "[";
        for (idx_t i = 0; i < e.length; i++) {
          o << (i ? "," : "");
          if (std::isnan(v[e.offset + i])) o << "NaN";      // what Python's json reads
          else o << std::setprecision(9) << v[e.offset + i];
        }
        o << "]";
        return o.str();
      };
      AggRun run(db.aggregates.at("sum_to_triple_3_3"), 1);
      auto cols = Cols("abcdef", all, false);
      run.Update(cols, {0, 0, 0, 0, 0});
      Vector triple = run.Finalize();
      DataChunk lt;      // linreg_train(triple, 0, 0.001, 0, 200, true, false)
      lt.data = {triple, constant(LogicalTypeId::INTEGER, 0), constant(LogicalTypeId::FLOAT, 0.001),
                 constant(LogicalTypeId::FLOAT, 0), constant(LogicalTypeId::INTEGER, 200),
                 constant(LogicalTypeId::BOOLEAN, 1), constant(LogicalTypeId::BOOLEAN, 0)};
      Vector lparams = call("linreg_train", lt, 1);
      out << ",\"linreg_params\":" << floats(lparams);
      DataChunk lp;      // linreg_predict(params, false, false, b, c, d, e, f) — dictionary vectors
      lp.data = {lparams, constant(LogicalTypeId::BOOLEAN, 0), constant(LogicalTypeId::BOOLEAN, 0)};
      for (auto &c : Cols("bcdef", all, true)) lp.data.push_back(c);
      Vector lpred = call("linreg_predict", lp, 5);
      out << ",\"linreg_pred\":[";
      for (int i = 0; i < 5; i++) out << (i ? "," : "") << std::setprecision(9) << FlatVector::GetData<float>(lpred)[i];
      out << "]";
      DataChunk dt;      // lda_train(triple, 0, 0.1, false): class = column d
      dt.data = {triple, constant(LogicalTypeId::INTEGER, 0), constant(LogicalTypeId::FLOAT, 0.1),
                 constant(LogicalTypeId::BOOLEAN, 0)};
      Vector dparams = call("lda_train", dt, 1);
      out << ",\"lda_params\":" << floats(dparams);
      DataChunk dp;      // lda_predict(params, false, a, b, c, e, f)
      dp.data = {dparams, constant(LogicalTypeId::BOOLEAN, 0)};
      for (auto &c : Cols("abcef", all, false)) dp.data.push_back(c);
      Vector dpred = call("lda_predict", dp, 5);
      out << ",\"lda_pred\":[";
      for (int i = 0; i < 5; i++) out << (i ? "," : "") << FlatVector::GetData<int32_t>(dpred)[i];
      out << "]";
    }
    out << "}";
    printf("%s\n", out.str().c_str());
  } catch (const std::exception &e) {
    fprintf(stderr, "glue_driver: %s\n", e.what());
    return 1;
  }
  return 0;
}
