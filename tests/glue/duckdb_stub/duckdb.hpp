// =============================================================================================
// TEST STAND-IN for the parts of the DuckDB v0.9.2 C++ API that OUR glue
// (duckdb-imputation_amd/duckdb_extension/src/duckdb_imputation_extension.cpp) touches.
//
// The image has no DuckDB.  This header exists so that the glue can be compiled and its callbacks
// driven by tests/glue/glue_driver.cpp (registration, bind, update with selection vectors and
// per-row state pointers, combine, finalize into nested LIST/STRUCT vectors, scalar functions on
// DataChunks).  It is NOT used to build anything of the reference, is not part of the product,
// and only models containers: vectors own flat buffers, LIST vectors own one child vector and a
// size, STRUCT vectors own their entries, a dictionary vector is a flat buffer plus a selection.
// Names and signatures follow DuckDB 0.9.2 as the reference uses them
// (duckdb_extension/src/duckdb_imputation_extension.cpp:48-113, triple/sum/sum_state.cpp).
// =============================================================================================
#pragma once
#include <cassert>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#define DUCKDB_EXTENSION_API
#define D_ASSERT(x) assert(x)

namespace duckdb {

using idx_t = uint64_t;
using sel_t = uint32_t;
using data_t = uint8_t;
using data_ptr_t = data_t *;
using std::string;
template <class T> using vector = std::vector<T>;
template <class T> using unique_ptr = std::unique_ptr<T>;
template <class T> using shared_ptr = std::shared_ptr<T>;
template <class T, class... A> unique_ptr<T> make_uniq(A &&...a) { return unique_ptr<T>(new T(std::forward<A>(a)...)); }
template <class T> using child_list_t = std::vector<std::pair<std::string, T>>;

struct list_entry_t { uint64_t offset; uint64_t length; };

static inline std::string FormatMessage(const char *fmt, va_list ap) {
  char buf[1024];
  vsnprintf(buf, sizeof(buf), fmt, ap);
  return buf;
}
struct Exception : std::runtime_error { using std::runtime_error::runtime_error; };
struct IOException : Exception {
  explicit IOException(const char *fmt, ...) : Exception(Make(fmt, nullptr)) {}
  template <class... A> IOException(const char *fmt, A... a) : Exception(Fmt(fmt, a...)) {}
  static std::string Make(const char *f, void *) { return f; }
  static std::string Fmt(const char *fmt, ...) { va_list ap; va_start(ap, fmt); auto s = FormatMessage(fmt, ap); va_end(ap); return s; }
};
struct InvalidInputException : Exception {
  template <class... A> InvalidInputException(const char *fmt, A... a) : Exception(IOException::Fmt(fmt, a...)) {}
};

enum class LogicalTypeId : uint8_t { INVALID, ANY, BOOLEAN, INTEGER, FLOAT, DOUBLE, VARCHAR, LIST, STRUCT };
enum class PhysicalType : uint8_t { INVALID, INT32, FLOAT, DOUBLE, VARCHAR, LIST, STRUCT };
enum class VectorType : uint8_t { FLAT_VECTOR, CONSTANT_VECTOR, DICTIONARY_VECTOR };
enum class FunctionNullHandling : uint8_t { DEFAULT_NULL_HANDLING, SPECIAL_HANDLING };

struct LogicalType {
  LogicalTypeId id_ = LogicalTypeId::INVALID;
  std::shared_ptr<child_list_t<LogicalType>> children;   // STRUCT fields, or one unnamed child for LIST
  LogicalType() {}
  LogicalType(LogicalTypeId id) : id_(id) {}               // NOLINT (implicit, as in DuckDB)
  LogicalTypeId id() const { return id_; }
  PhysicalType InternalType() const {
    switch (id_) {
    case LogicalTypeId::INTEGER: return PhysicalType::INT32;
    case LogicalTypeId::FLOAT: return PhysicalType::FLOAT;
    case LogicalTypeId::DOUBLE: return PhysicalType::DOUBLE;
    case LogicalTypeId::LIST: return PhysicalType::LIST;
    case LogicalTypeId::STRUCT: return PhysicalType::STRUCT;
    default: return PhysicalType::INVALID;
    }
  }
  bool operator==(const LogicalType &o) const { return id_ == o.id_; }
  bool operator!=(const LogicalType &o) const { return id_ != o.id_; }
  static LogicalType LIST(const LogicalType &child) {
    LogicalType t(LogicalTypeId::LIST);
    t.children = std::make_shared<child_list_t<LogicalType>>();
    t.children->emplace_back("", child);
    return t;
  }
  static LogicalType STRUCT(child_list_t<LogicalType> fields) {
    LogicalType t(LogicalTypeId::STRUCT);
    t.children = std::make_shared<child_list_t<LogicalType>>(std::move(fields));
    return t;
  }
  static constexpr LogicalTypeId BOOLEAN = LogicalTypeId::BOOLEAN;
  static constexpr LogicalTypeId INTEGER = LogicalTypeId::INTEGER;
  static constexpr LogicalTypeId FLOAT = LogicalTypeId::FLOAT;
  static constexpr LogicalTypeId DOUBLE = LogicalTypeId::DOUBLE;
  static constexpr LogicalTypeId VARCHAR = LogicalTypeId::VARCHAR;
  static constexpr LogicalTypeId ANY = LogicalTypeId::ANY;
};
inline bool operator==(const LogicalType &a, LogicalTypeId b) { return a.id() == b; }
inline bool operator!=(const LogicalType &a, LogicalTypeId b) { return a.id() != b; }
struct ListType { static const LogicalType &GetChildType(const LogicalType &t) { return (*t.children)[0].second; } };
struct StructType { static const child_list_t<LogicalType> &GetChildTypes(const LogicalType &t) { return *t.children; } };

static inline idx_t TypeSize(const LogicalType &t) {
  switch (t.id()) {
  case LogicalTypeId::BOOLEAN: return 1;
  case LogicalTypeId::INTEGER: case LogicalTypeId::FLOAT: return 4;
  case LogicalTypeId::DOUBLE: return 8;
  case LogicalTypeId::LIST: return sizeof(list_entry_t);
  default: return 8;   // pointers (state vectors)
  }
}

struct SelectionVector {
  sel_t *sel_vector = nullptr;
  SelectionVector() {}
  explicit SelectionVector(sel_t *s) : sel_vector(s) {}
  idx_t get_index(idx_t i) const { return sel_vector ? sel_vector[i] : i; }
  sel_t *data() { return sel_vector; }
  const sel_t *data() const { return sel_vector; }
};

struct UnifiedVectorFormat {
  const SelectionVector *sel = nullptr;
  data_ptr_t data = nullptr;
  SelectionVector owned_sel;
  template <class T> static const T *GetData(const UnifiedVectorFormat &f) { return reinterpret_cast<const T *>(f.data); }
};

// what Vector::GetValue(i).GetValue<T>() needs: a scalar that casts
class Value {
public:
  explicit Value(double v) : d(v) {}
  template <class T> T GetValue() const { return static_cast<T>(d); }
private:
  double d;
};

class Vector {
public:
  explicit Vector(LogicalType t, idx_t capacity = 2048) : type(std::move(t)) { Init(capacity); }
  Vector(const Vector &o) = default;                        // shares buffers, like DuckDB vector references
  LogicalType &GetType() { return type; }
  const LogicalType &GetType() const { return type; }
  VectorType GetVectorType() const { return vtype; }
  void SetVectorType(VectorType v) { vtype = v; }
  // Dictionary vector for tests: logical row i reads physical row sel[i] of the (flat) buffer.
  void Slice(std::shared_ptr<std::vector<sel_t>> s) { dict = std::move(s); vtype = VectorType::DICTIONARY_VECTOR; }
  void ToUnifiedFormat(idx_t, UnifiedVectorFormat &f) {
    f.data = buffer->data();
    f.owned_sel = SelectionVector(dict ? dict->data() : nullptr);
    f.sel = &f.owned_sel;
  }
  void Flatten(idx_t count) {
    if (vtype == VectorType::FLAT_VECTOR || !dict) { vtype = VectorType::FLAT_VECTOR; return; }
    const idx_t w = TypeSize(type);
    auto nb = std::make_shared<std::vector<data_t>>(std::max<idx_t>(count, 1) * w);
    for (idx_t i = 0; i < count; i++) std::memcpy(nb->data() + i * w, buffer->data() + (*dict)[i] * w, w);
    buffer = nb; dict.reset(); vtype = VectorType::FLAT_VECTOR;
  }
  Value GetValue(idx_t i) const {
    const idx_t r = dict ? (*dict)[i] : i;
    switch (type.id()) {
    case LogicalTypeId::BOOLEAN: return Value(buffer->data()[r] != 0);
    case LogicalTypeId::INTEGER: return Value(reinterpret_cast<const int32_t *>(buffer->data())[r]);
    case LogicalTypeId::FLOAT: return Value(reinterpret_cast<const float *>(buffer->data())[r]);
    case LogicalTypeId::DOUBLE: return Value(reinterpret_cast<const double *>(buffer->data())[r]);
    default: throw InvalidInputException("stub: GetValue on a nested vector");
    }
  }
  // internals (public for the helper structs below)
  LogicalType type;
  VectorType vtype = VectorType::FLAT_VECTOR;
  std::shared_ptr<std::vector<data_t>> buffer;
  std::shared_ptr<std::vector<sel_t>> dict;
  std::shared_ptr<Vector> list_child;                       // LIST
  std::shared_ptr<idx_t> list_size;                         // LIST
  std::shared_ptr<vector<unique_ptr<Vector>>> entries;      // STRUCT
  void Resize(idx_t rows) {
    const idx_t need = std::max<idx_t>(rows, 1) * TypeSize(type);
    if (buffer->size() < need) buffer->resize(need);
    if (entries) for (auto &e : *entries) e->Resize(rows);
  }
private:
  void Init(idx_t capacity) {
    buffer = std::make_shared<std::vector<data_t>>(std::max<idx_t>(capacity, 1) * TypeSize(type));
    if (type.id() == LogicalTypeId::LIST) {
      list_child = std::make_shared<Vector>(ListType::GetChildType(type), capacity);
      list_size = std::make_shared<idx_t>(0);
    } else if (type.id() == LogicalTypeId::STRUCT) {
      entries = std::make_shared<vector<unique_ptr<Vector>>>();
      for (auto const &f : StructType::GetChildTypes(type)) entries->push_back(make_uniq<Vector>(f.second, capacity));
    }
  }
};

struct FlatVector {
  template <class T> static T *GetData(Vector &v) { return reinterpret_cast<T *>(v.buffer->data()); }
  static data_ptr_t GetData(Vector &v) { return v.buffer->data(); }
  static void SetNull(Vector &, idx_t, bool) {}
};
struct ListVector {
  static Vector &GetEntry(Vector &v) { return *v.list_child; }
  static list_entry_t *GetData(Vector &v) { return reinterpret_cast<list_entry_t *>(v.buffer->data()); }
  static idx_t GetListSize(Vector &v) { return *v.list_size; }
  static void SetListSize(Vector &v, idx_t n) { *v.list_size = n; }
  static void Reserve(Vector &v, idx_t n) { v.list_child->Resize(n); }      // may move the child buffers
};
struct StructVector {
  static vector<unique_ptr<Vector>> &GetEntries(Vector &v) { return *v.entries; }
};

class DataChunk {
public:
  vector<Vector> data;
  idx_t count = 0;
  idx_t size() const { return count; }
  idx_t ColumnCount() const { return data.size(); }
};

struct ClientContext {};
struct ExpressionState {};
struct Expression {};
struct FunctionData {
  virtual ~FunctionData() {}
  virtual unique_ptr<FunctionData> Copy() const = 0;
  virtual bool Equals(const FunctionData &other) const = 0;
  // DuckDB's Cast is an unchecked reinterpret in release builds; the stand-in checks, so a glue that
  // casts a plain VariableReturnBindData to its own subclass fails the test instead of reading garbage
  template <class TARGET> TARGET &Cast() { return dynamic_cast<TARGET &>(*this); }
  template <class TARGET> const TARGET &Cast() const { return dynamic_cast<const TARGET &>(*this); }
};
struct AggregateInputData { FunctionData *bind_data = nullptr; };   // (optional_ptr<FunctionData> in DuckDB)
struct VariableReturnBindData : FunctionData {
  LogicalType stype;
  explicit VariableReturnBindData(LogicalType t) : stype(std::move(t)) {}
  unique_ptr<FunctionData> Copy() const override { return make_uniq<VariableReturnBindData>(stype); }   // (as DuckDB 0.9.2's)
  bool Equals(const FunctionData &other_p) const override { return stype == other_p.Cast<VariableReturnBindData>().stype; }
  static void Serialize() {}
  static void Deserialize() {}
};

typedef idx_t (*aggregate_size_t)();
typedef void (*aggregate_initialize_t)(data_ptr_t state);
typedef void (*aggregate_update_t)(Vector inputs[], AggregateInputData &, idx_t input_count, Vector &state, idx_t count);
typedef void (*aggregate_combine_t)(Vector &state, Vector &combined, AggregateInputData &, idx_t count);
typedef void (*aggregate_finalize_t)(Vector &state, AggregateInputData &, Vector &result, idx_t count, idx_t offset);
typedef void (*aggregate_destructor_t)(Vector &state, AggregateInputData &, idx_t count);
class AggregateFunction;
typedef unique_ptr<FunctionData> (*bind_aggregate_function_t)(ClientContext &, AggregateFunction &, vector<unique_ptr<Expression>> &);

class AggregateFunction {
public:
  AggregateFunction(const string &name, const vector<LogicalType> &arguments, const LogicalType &return_type,
                    aggregate_size_t state_size, aggregate_initialize_t initialize, aggregate_update_t update,
                    aggregate_combine_t combine, aggregate_finalize_t finalize, void *simple_update = nullptr,
                    bind_aggregate_function_t bind = nullptr, aggregate_destructor_t destructor = nullptr,
                    void *statistics = nullptr, void *window = nullptr)
      : name(name), arguments(arguments), return_type(return_type), state_size(state_size), initialize(initialize),
        update(update), combine(combine), finalize(finalize), bind(bind), destructor(destructor) {
    (void)simple_update; (void)statistics; (void)window;
  }
  string name;
  vector<LogicalType> arguments;
  LogicalType return_type;
  LogicalType varargs;
  FunctionNullHandling null_handling = FunctionNullHandling::DEFAULT_NULL_HANDLING;
  aggregate_size_t state_size;
  aggregate_initialize_t initialize;
  aggregate_update_t update;
  aggregate_combine_t combine;
  aggregate_finalize_t finalize;
  bind_aggregate_function_t bind;
  aggregate_destructor_t destructor;

  template <class STATE> static idx_t StateSize() { return sizeof(STATE); }
  template <class STATE, class OP> static void StateInitialize(data_ptr_t state) { OP::Initialize(*reinterpret_cast<STATE *>(state)); }
  template <class STATE, class OP> static void StateDestroy(Vector &states, AggregateInputData &aid, idx_t count) {
    auto sdata = FlatVector::GetData<STATE *>(states);
    for (idx_t i = 0; i < count; i++) OP::template Destroy<STATE>(*sdata[i], aid);
  }
};

class ScalarFunction;
typedef void (*scalar_function_t)(DataChunk &, ExpressionState &, Vector &);
typedef unique_ptr<FunctionData> (*bind_scalar_function_t)(ClientContext &, ScalarFunction &, vector<unique_ptr<Expression>> &);
class ScalarFunction {
public:
  ScalarFunction(const string &name, const vector<LogicalType> &arguments, const LogicalType &return_type,
                 scalar_function_t function, bind_scalar_function_t bind = nullptr, void *dependency = nullptr,
                 void *statistics = nullptr)
      : name(name), arguments(arguments), return_type(return_type), function(function), bind(bind) {
    (void)dependency; (void)statistics;
  }
  string name;
  vector<LogicalType> arguments;
  LogicalType return_type;
  LogicalType varargs;
  FunctionNullHandling null_handling = FunctionNullHandling::DEFAULT_NULL_HANDLING;
  scalar_function_t function;
  bind_scalar_function_t bind;
  void (*serialize)() = nullptr;
  void (*deserialize)() = nullptr;
};

class DatabaseInstance {
public:
  std::map<string, AggregateFunction> aggregates;
  std::map<string, ScalarFunction> scalars;
};
struct ExtensionUtil {
  static void RegisterFunction(DatabaseInstance &db, AggregateFunction f) { db.aggregates.emplace(f.name, std::move(f)); }
  static void RegisterFunction(DatabaseInstance &db, ScalarFunction f) { db.scalars.emplace(f.name, std::move(f)); }
};

class DuckDB;
class Extension {
public:
  virtual ~Extension() {}
  virtual void Load(DuckDB &db) = 0;
  virtual std::string Name() = 0;
};
class DuckDB {
public:
  explicit DuckDB(DatabaseInstance &db) : instance(&db) {}
  DatabaseInstance *instance;
  template <class T> void LoadExtension() { T ext; ext.Load(*this); }
  static const char *LibraryVersion() { return "v0.9.2"; }
};

}  // namespace duckdb
