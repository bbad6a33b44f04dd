// =============================================================================================
// oracle/cofactor_oracle.cpp — CPU restatement of the reference's cofactor-triple ring.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under duckdb-imputation_amd/ (the product) may include,
// link, load or call this file.  Only tests/, __graft_entry__.smoke() and bench.py's
// `cpu_baseline` leg use it, and only as the checker / reported baseline.
//
// What it is: a from-scratch C++ restatement of the reference algorithm, function by
// function, with the same accumulation order per accumulator (so that in `faithful` mode —
// float accumulators, int32 N — it is arithmetically the reference; in `wide` mode — double
// accumulators, int64 N — it is the truth the GPU path is held to at 1e-5 relative).
// std::map is kept for the categorical tables because (a) the finalised key order IS the
// std::map iteration order in the reference and (b) the CPU baseline should pay what the
// reference pays.
//
// Pinning: the reference needs <duckdb.hpp> (DuckDB v0.9.2, not vendored, not
// in this image), so the reference itself is unbuildable here; this restatement is pinned by
// the reference's own known-answer tests instead — every literal in
// duckdb_extension/test/python/test_{sum,lift,mul,nb_sum,nb_lift,nb_mul}.py, extracted into
// tests/golden/ring_goldens.json by tests/golden/make_golden.py and asserted in
// tests/test_oracle_golden.py.  A12 subtract has no reference test ("parity unpinned").
//
// Flat triple blob (array of doubles; every int32/float value is exactly representable), in the
// order the reference's finalize writes its nested vectors (sum_state.cpp:116-464):
//   [0] kind (0 = triple, 1 = nb)   [1] n   [2] m   [3] N
//   lin[n]
//   quad[T(n)]   (kind 0, row-major upper triangle)   |   quad[n]  (kind 1, diagonal)
//   lin_cat:       m lists, each:  len, len x (key, value)
//   quad_num_cat:  n*m lists (index k*m + c), each: len, len x (key, value)       (kind 0 only)
//   quad_cat:      T(m) lists (c1 outer, c2 >= c1), each: len, len x (k1, k2, value) (kind 0 only)
// =============================================================================================
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

namespace orc {

static inline int64_t tri(int64_t k) { return k * (k + 1) / 2; }

// ---- aggregate state: restates Triple::SumState (include/triple/sum/sum_state.h:14-28) -------
template <typename F, typename C>
struct State {
  C count = 0;
  int n = 0, m = 0;
  bool nb = false;
  bool allocated = false;                                   // lin_agg/quad_num_cat != nullptr
  std::vector<F> lin, quad;                                 // lin_agg, quadratic_agg
  std::vector<std::map<int, std::vector<F>>> num_cat;       // per cat col: key -> [cnt, Sx_0..]
  std::vector<std::map<std::pair<int, int>, F>> cat_cat;    // per cat pair: (k1,k2) -> cnt

  void alloc(int n_, int m_, bool nb_) {                    // sum_no_lift.cpp:96-116,
    n = n_; m = m_; nb = nb_; allocated = true;             // sum_to_nb_agg.cpp:74-90
    lin.assign(n, F(0));
    quad.assign(nb ? n : tri(n), F(0));
    num_cat.assign(m, {});
    cat_cat.assign(nb ? 0 : tri(m), {});
  }
};

// ---- update: Triple::SumNoLift (sum_no_lift.cpp:53-216) and sum_to_nb_agg
//      (sum_to_nb_agg.cpp:39-146).  `group[i]` plays the role of the per-row state pointer
//      (state_vector), nullptr = every row goes to states[0].  One call = one DataChunk.
//      gbase: group is indexed with (i - gbase) — a per-CHUNK pointer vector of 2048 entries, as
//      DuckDB hands one state vector per chunk (sum_no_lift.cpp:84,94,139 read states[sel[i]] per row).
template <typename F, typename C>
static void update_chunk(State<F, C>** states, const int32_t* group, const float* const* num,
                         int n, const int32_t* const* cat, int m, int64_t lo, int64_t hi,
                         bool nb, int64_t gbase = 0) {
  auto st = [&](int64_t i) -> State<F, C>& { return *states[group ? group[i - gbase] : 0]; };
  for (int64_t i = lo; i < hi; i++) st(i).count += 1;                       // :83-86
  for (int64_t i = lo; i < hi; i++) {                                       // :90-122
    auto& s = st(i);
    if (!s.allocated) s.alloc(n, m, nb);
    for (int k = 0; k < n; k++) s.lin[k] += num[k][i];
  }
  if (!nb) {
    int q = 0;                                                              // :128-146
    for (int j = 0; j < n; j++)
      for (int k = j; k < n; k++, q++) {
        const float* a = num[j];
        const float* b = num[k];
        for (int64_t i = lo; i < hi; i++) st(i).quad[q] += a[i] * b[i];     // float product
      }
  } else {
    for (int j = 0; j < n; j++) {                                           // nb :107-117
      const float* a = num[j];
      for (int64_t i = lo; i < hi; i++) st(i).quad[j] += a[i] * a[i];
    }
  }
  if (m > 0) {                                                              // :157-189
    for (int64_t i = lo; i < hi; i++) {
      auto& s = st(i);
      for (int c = 0; c < m; c++) {
        auto& tab = s.num_cat[c];
        int key = cat[c][i];
        auto pos = tab.find(key);
        if (pos == tab.end()) {
          std::vector<F> payload(nb ? 1 : n + 1);
          payload[0] = F(1);
          if (!nb) for (int k = 0; k < n; k++) payload[k + 1] = num[k][i];
          tab.emplace(key, std::move(payload));
        } else {
          auto& payload = pos->second;
          if (!nb) for (int k = 0; k < n; k++) payload[k + 1] += num[k][i];
          payload[0] += F(1);
        }
      }
    }
  }
  if (!nb) {
    int q = 0;                                                              // :195-214
    for (int c1 = 0; c1 < m; c1++)
      for (int c2 = c1; c2 < m; c2++, q++)
        for (int64_t i = lo; i < hi; i++) {
          auto& tab = st(i).cat_cat[q];
          std::pair<int, int> key(cat[c1][i], cat[c2][i]);
          auto pos = tab.find(key);
          if (pos == tab.end()) tab.emplace(key, F(1));
          else pos->second += F(1);
        }
  }
}

// ---- combine: Triple::SumStateCombine (sum_state.cpp:10-114) ---------------------------------
template <typename F, typename C>
static void combine(State<F, C>& dst, const State<F, C>& src) {
  dst.count += src.count;                                                   // :25
  if (!dst.allocated) {                                                     // :26-60
    if (!src.allocated) { dst.nb = src.nb; return; }                        // empty += empty
    dst.alloc(src.n, src.m, src.nb);
  }
  if (!src.allocated) return;  // reference would dereference null here; an empty source adds 0
  for (int j = 0; j < dst.n; j++) dst.lin[j] += src.lin[j];                 // :73-75
  for (size_t j = 0; j < dst.quad.size(); j++) dst.quad[j] += src.quad[j];  // :76-83
  for (int c = 0; c < dst.m; c++) {                                         // :87-97
    auto& tab = dst.num_cat[c];
    for (auto const& kv : src.num_cat[c]) {
      auto pos = tab.find(kv.first);
      if (pos == tab.end()) tab.emplace(kv.first, kv.second);
      else for (size_t k = 0; k < kv.second.size(); k++) pos->second[k] += kv.second[k];
    }
  }
  if (!src.nb)                                                              // :100-111
    for (size_t q = 0; q < dst.cat_cat.size(); q++) {
      auto& tab = dst.cat_cat[q];
      for (auto const& kv : src.cat_cat[q]) {
        auto pos = tab.find(kv.first);
        if (pos == tab.end()) tab.emplace(kv.first, kv.second);
        else pos->second += kv.second;
      }
    }
}

// ---- finalize: Triple::SumStateFinalize (sum_state.cpp:116-464) -> flat blob -------------------
template <typename F, typename C>
static void finalize(const State<F, C>& s, std::vector<double>& out) {
  out.clear();
  out.push_back(s.nb ? 1 : 0);
  out.push_back(s.n);
  out.push_back(s.m);
  out.push_back((double)s.count);                                           // :132-135
  for (int k = 0; k < s.n; k++) out.push_back(s.allocated ? (double)s.lin[k] : 0.0);  // :162-171
  size_t qn = s.nb ? s.n : tri(s.n);
  for (size_t k = 0; k < qn; k++) out.push_back(s.allocated ? (double)s.quad[k] : 0.0);  // :177-200
  for (int c = 0; c < s.m; c++) {                                           // lin_cat :372-395
    out.push_back((double)s.num_cat[c].size());
    for (auto const& kv : s.num_cat[c]) { out.push_back(kv.first); out.push_back((double)kv.second[0]); }
  }
  if (s.nb) return;                                                         // :300-325
  for (int k = 0; k < s.n; k++)                                             // quad_num_cat :384,:397-404
    for (int c = 0; c < s.m; c++) {
      out.push_back((double)s.num_cat[c].size());
      for (auto const& kv : s.num_cat[c]) { out.push_back(kv.first); out.push_back((double)kv.second[k + 1]); }
    }
  for (size_t q = 0; q < s.cat_cat.size(); q++) {                           // quad_cat :440-461
    out.push_back((double)s.cat_cat[q].size());
    for (auto const& kv : s.cat_cat[q]) {
      out.push_back(kv.first.first); out.push_back(kv.first.second); out.push_back((double)kv.second);
    }
  }
}

// ---- a finalised triple as a value (what DuckDB hands to sum_triple / multiply_triple) ---------
struct Triple {
  int kind = 0, n = 0, m = 0;
  double N = 0;
  std::vector<double> lin, quad;
  std::vector<std::vector<std::pair<int, double>>> lin_cat, num_cat;
  std::vector<std::vector<std::tuple<int, int, double>>> cat_cat;
};

static const double* parse(const double* p, Triple& t) {
  t.kind = (int)p[0]; t.n = (int)p[1]; t.m = (int)p[2]; t.N = p[3];
  p += 4;
  t.lin.assign(p, p + t.n); p += t.n;
  size_t qn = t.kind ? t.n : tri(t.n);
  t.quad.assign(p, p + qn); p += qn;
  auto read_kv = [&](std::vector<std::vector<std::pair<int, double>>>& dst, int lists) {
    dst.assign(lists, {});
    for (int l = 0; l < lists; l++) {
      int len = (int)*p++;
      for (int e = 0; e < len; e++, p += 2) dst[l].emplace_back((int)p[0], p[1]);
    }
  };
  read_kv(t.lin_cat, t.m);
  t.num_cat.clear(); t.cat_cat.clear();
  if (t.kind == 0) {
    read_kv(t.num_cat, t.n * t.m);
    t.cat_cat.assign(tri(t.m), {});
    for (auto& lst : t.cat_cat) {
      int len = (int)*p++;
      for (int e = 0; e < len; e++, p += 3) lst.emplace_back((int)p[0], (int)p[1], p[2]);
    }
  }
  return p;
}

static void emit(const Triple& t, std::vector<double>& out) {
  out.push_back(t.kind); out.push_back(t.n); out.push_back(t.m); out.push_back(t.N);
  out.insert(out.end(), t.lin.begin(), t.lin.end());
  out.insert(out.end(), t.quad.begin(), t.quad.end());
  for (auto const& l : t.lin_cat) {
    out.push_back((double)l.size());
    for (auto const& kv : l) { out.push_back(kv.first); out.push_back(kv.second); }
  }
  if (t.kind) return;
  for (auto const& l : t.num_cat) {
    out.push_back((double)l.size());
    for (auto const& kv : l) { out.push_back(kv.first); out.push_back(kv.second); }
  }
  for (auto const& l : t.cat_cat) {
    out.push_back((double)l.size());
    for (auto const& e : l) { out.push_back(std::get<0>(e)); out.push_back(std::get<1>(e)); out.push_back(std::get<2>(e)); }
  }
}

// ---- lift: Triple::CustomLift (lift.cpp:15-243) and to_nb_lift (lift_to_nb_agg.cpp:13-136) ----
static void lift_row(const float* const* num, int n, const int32_t* const* cat, int m,
                     int64_t i, bool nb, Triple& t) {
  t.kind = nb; t.n = n; t.m = m; t.N = 1;                                   // :44-46
  t.lin.resize(n);
  for (int k = 0; k < n; k++) t.lin[k] = num[k][i];                         // :85-93
  t.lin_cat.assign(m, {});
  for (int c = 0; c < m; c++) t.lin_cat[c].emplace_back(cat[c][i], 1.0);    // :94-105
  t.quad.clear();
  if (nb) {
    for (int j = 0; j < n; j++) t.quad.push_back((double)(num[j][i] * num[j][i]));
    return;
  }
  for (int j = 0; j < n; j++)                                               // :119-136
    for (int k = j; k < n; k++) t.quad.push_back((double)(num[j][i] * num[k][i]));
  t.num_cat.assign((size_t)n * m, {});                                      // :156-176
  for (int j = 0; j < n; j++)
    for (int c = 0; c < m; c++) t.num_cat[(size_t)j * m + c].emplace_back(cat[c][i], (double)num[j][i]);
  t.cat_cat.assign(tri(m), {});                                             // :199-219
  int q = 0;
  for (int c1 = 0; c1 < m; c1++)
    for (int c2 = c1; c2 < m; c2++, q++) t.cat_cat[q].emplace_back(cat[c1][i], cat[c2][i], 1.0);
}

// ---- sum of already-lifted triples: Triple::Sum (sum.cpp:57-261), sum_nb_agg
//      (sum_nb_agg.cpp:45-175).  One call adds one input triple into the state. ----------------
template <typename F, typename C>
static void sum_one(State<F, C>& s, const Triple& t) {
  s.count += (C)t.N;                                                        // :86-89
  if (!s.allocated) s.alloc(t.n, t.m, t.kind != 0);                         // :110-127
  for (int k = 0; k < t.n; k++) s.lin[k] += (F)t.lin[k];                    // :129-131
  for (size_t k = 0; k < t.quad.size(); k++) s.quad[k] += (F)t.quad[k];     // :142-149
  for (int c = 0; c < t.m; c++) {                                           // :197-226
    auto& tab = s.num_cat[c];
    auto const& keys = t.lin_cat[c];
    for (size_t e = 0; e < keys.size(); e++) {
      std::vector<F> vals(t.kind ? 1 : t.n + 1);
      vals[0] = (F)keys[e].second;
      if (!t.kind)
        for (int l = 0; l < t.n; l++) vals[l + 1] = (F)t.num_cat[(size_t)l * t.m + c][e].second;
      auto pos = tab.find(keys[e].first);
      if (pos == tab.end()) tab.emplace(keys[e].first, std::move(vals));
      else for (size_t l = 0; l < vals.size(); l++) pos->second[l] += vals[l];
    }
  }
  if (!t.kind)
    for (size_t q = 0; q < t.cat_cat.size(); q++) {                         // :246-260
      auto& tab = s.cat_cat[q];
      for (auto const& e : t.cat_cat[q]) {
        std::pair<int, int> key(std::get<0>(e), std::get<1>(e));
        auto pos = tab.find(key);
        if (pos == tab.end()) tab.emplace(key, (F)std::get<2>(e));
        else pos->second += (F)std::get<2>(e);
      }
    }
}

// ---- ring product: Triple::MultiplyFunction (mul.cpp:19-611), multiply_nb (mul_nb.cpp:20-268)
template <typename F>
static void multiply(const Triple& A, const Triple& B, Triple& R) {
  const bool nb = A.kind != 0;
  R = Triple();
  R.kind = A.kind; R.n = A.n + B.n; R.m = A.m + B.m;
  const int32_t NA = (int32_t)A.N, NB = (int32_t)B.N;
  R.N = (double)(int32_t)(NA * NB);                                         // :46-49 int32 product
  auto mulF = [](double x, double y) { return (double)((F)x * (F)y); };
  for (int j = 0; j < A.n; j++) R.lin.push_back(mulF(A.lin[j], NB));        // :97-107
  for (int j = 0; j < B.n; j++) R.lin.push_back(mulF(B.lin[j], NA));
  auto scaled = [&](const std::vector<std::pair<int, double>>& l, double f) {
    std::vector<std::pair<int, double>> o;
    for (auto const& kv : l) o.emplace_back(kv.first, mulF(kv.second, f));
    return o;
  };
  for (int c = 0; c < A.m; c++) R.lin_cat.push_back(scaled(A.lin_cat[c], NB));  // :185-217
  for (int c = 0; c < B.m; c++) R.lin_cat.push_back(scaled(B.lin_cat[c], NA));
  if (nb) {                                                                 // mul_nb.cpp:246-262
    for (int j = 0; j < A.n; j++) R.quad.push_back(mulF(A.quad[j], NB));
    for (int j = 0; j < B.n; j++) R.quad.push_back(mulF(B.quad[j], NA));
    return;
  }
  int q = 0;                                                                // :262-289
  for (int j = 0; j < A.n; j++) {
    for (int k = j; k < A.n; k++) R.quad.push_back(mulF(A.quad[q++], NB));
    for (int k = 0; k < B.n; k++) R.quad.push_back(mulF(A.lin[j], B.lin[k]));
  }
  for (size_t k = 0; k < B.quad.size(); k++) R.quad.push_back(mulF(B.quad[k], NA));
  for (int j = 0; j < A.n; j++) {                                           // :377-446
    for (int c = 0; c < A.m; c++) R.num_cat.push_back(scaled(A.num_cat[(size_t)j * A.m + c], NB));
    for (int c = 0; c < B.m; c++) R.num_cat.push_back(scaled(B.lin_cat[c], A.lin[j]));
  }
  for (int j = 0; j < B.n; j++) {
    for (int c = 0; c < A.m; c++) R.num_cat.push_back(scaled(A.lin_cat[c], B.lin[j]));
    for (int c = 0; c < B.m; c++) R.num_cat.push_back(scaled(B.num_cat[(size_t)j * B.m + c], NA));
  }
  q = 0;                                                                    // :542-598
  for (int c1 = 0; c1 < A.m; c1++) {
    for (int c2 = c1; c2 < A.m; c2++, q++) {
      std::vector<std::tuple<int, int, double>> o;
      for (auto const& e : A.cat_cat[q]) o.emplace_back(std::get<0>(e), std::get<1>(e), mulF(std::get<2>(e), NB));
      R.cat_cat.push_back(std::move(o));
    }
    for (int c2 = 0; c2 < B.m; c2++) {                                      // key-set outer product :564-580
      std::vector<std::tuple<int, int, double>> o;
      for (auto const& ka : A.lin_cat[c1])
        for (auto const& kb : B.lin_cat[c2]) o.emplace_back(ka.first, kb.first, mulF(ka.second, kb.second));
      R.cat_cat.push_back(std::move(o));
    }
  }
  for (size_t k = 0; k < B.cat_cat.size(); k++) {
    std::vector<std::tuple<int, int, double>> o;
    for (auto const& e : B.cat_cat[k]) o.emplace_back(std::get<0>(e), std::get<1>(e), mulF(std::get<2>(e), NA));
    R.cat_cat.push_back(std::move(o));
  }
}

// ---- host Value-level t1 (+|-) t2: Triple::sum_triple (imputation/triple/sum.cpp:68-209),
//      subtract_triple (imputation/triple/sub.cpp:71-217), sum_nb_triple (sum_nb.cpp:38-83) ----
template <typename F>
static void add_sub(const Triple& A, const Triple& B, bool sub, Triple& R) {
  R = Triple();
  R.kind = A.kind; R.n = std::max(A.n, B.n); R.m = std::max(A.m, B.m);
  R.N = sub ? A.N - B.N : A.N + B.N;
  auto op = [&](double x, double y) { return sub ? (double)((F)x - (F)y) : (double)((F)x + (F)y); };
  auto dense = [&](const std::vector<double>& a, const std::vector<double>& b, std::vector<double>& r) {
    if (!a.empty() && !b.empty()) for (size_t i = 0; i < a.size(); i++) r.push_back(op(a[i], b[i]));
    else if (!a.empty()) r = a;                                             // sum.cpp:82-93
    else if (!b.empty()) r = b;
  };
  dense(A.lin, B.lin, R.lin);
  dense(A.quad, B.quad, R.quad);
  auto kv = [&](const std::vector<std::vector<std::pair<int, double>>>& a,
                const std::vector<std::vector<std::pair<int, double>>>& b,
                std::vector<std::vector<std::pair<int, double>>>& r) {
    if (a.empty()) { r = b; return; }
    if (b.empty()) { r = a; return; }
    for (size_t l = 0; l < a.size(); l++) {                                 // sum.cpp:12-41 / sub.cpp:15-40
      std::map<int, F> content;
      for (auto const& e : a[l]) content[e.first] = (F)e.second;
      for (auto const& e : b[l]) {
        auto pos = content.find(e.first);
        if (pos == content.end()) { if (!sub) content[e.first] = (F)e.second; }  // sub: "Error, key is not present"
        else pos->second = sub ? pos->second - (F)e.second : pos->second + (F)e.second;
      }
      r.emplace_back();
      for (auto const& e : content) r.back().emplace_back(e.first, (double)e.second);
    }
  };
  kv(A.lin_cat, B.lin_cat, R.lin_cat);
  if (A.kind) return;
  kv(A.num_cat, B.num_cat, R.num_cat);
  if (A.cat_cat.empty()) { R.cat_cat = B.cat_cat; return; }
  if (B.cat_cat.empty()) { R.cat_cat = A.cat_cat; return; }
  for (size_t l = 0; l < A.cat_cat.size(); l++) {                           // sum.cpp:43-66 / sub.cpp:42-69
    std::map<std::pair<int, int>, F> content;
    for (auto const& e : A.cat_cat[l]) content[{std::get<0>(e), std::get<1>(e)}] = (F)std::get<2>(e);
    for (auto const& e : B.cat_cat[l]) {
      std::pair<int, int> key(std::get<0>(e), std::get<1>(e));
      auto pos = content.find(key);
      if (pos == content.end()) { if (!sub) content[key] = (F)std::get<2>(e); }
      else pos->second = sub ? pos->second - (F)std::get<2>(e) : pos->second + (F)std::get<2>(e);
    }
    R.cat_cat.emplace_back();
    for (auto const& e : content) R.cat_cat.back().emplace_back(e.first.first, e.first.second, (double)e.second);
  }
}

struct Handle {
  int mode;                                   // 0 = faithful (float,int32), 1 = wide (double,int64)
  State<float, int32_t> f;
  State<double, int64_t> d;
};

static const int64_t CHUNK = 2048;            // DuckDB STANDARD_VECTOR_SIZE

template <typename F, typename C>
static void run_update(State<F, C>** states, const int32_t* group, const float* const* num, int n,
                       const int32_t* const* cat, int m, int64_t lo, int64_t hi, bool nb) {
  for (int64_t a = lo; a < hi; a += CHUNK)
    update_chunk<F, C>(states, group, num, n, cat, m, a, std::min(hi, a + CHUNK), nb);
}

// The ungrouped aggregate exactly as the executor drives the reference: every chunk comes with a
// vector of 2048 state pointers (all the same state), read per row.  `chunk_sel` is that vector as
// indices into `states`; the compiler cannot know its entries are equal, so the accumulator loops
// keep the reference's per-row indirection instead of turning into plain vector sums
// ("per-row state-pointer indirection").
template <typename F, typename C>
static void run_update_ptr(State<F, C>** states, const int32_t* chunk_sel, const float* const* num, int n,
                           const int32_t* const* cat, int m, int64_t lo, int64_t hi, bool nb) {
  for (int64_t a = lo; a < hi; a += CHUNK)
    update_chunk<F, C>(states, chunk_sel, num, n, cat, m, a, std::min(hi, a + CHUNK), nb, /*gbase=*/a);
}

}  // namespace orc

using orc::Handle;

extern "C" {

void* orc_state_new(int mode) { auto* h = new Handle(); h->mode = mode; return h; }
void orc_state_free(void* p) { delete (Handle*)p; }

// Rows [0,rows) of the given columns, 2048-row chunks, group[i] selects states[group[i]].
int orc_update(void** states, int nstates, const int32_t* group, const float* const* num, int n,
               const int32_t* const* cat, int m, int64_t rows, int nb) {
  if (nstates < 1) return 1;
  int mode = ((Handle*)states[0])->mode;
  if (mode == 0) {
    std::vector<orc::State<float, int32_t>*> s(nstates);
    for (int i = 0; i < nstates; i++) s[i] = &((Handle*)states[i])->f;
    orc::run_update<float, int32_t>(s.data(), group, num, n, cat, m, 0, rows, nb != 0);
  } else {
    std::vector<orc::State<double, int64_t>*> s(nstates);
    for (int i = 0; i < nstates; i++) s[i] = &((Handle*)states[i])->d;
    orc::run_update<double, int64_t>(s.data(), group, num, n, cat, m, 0, rows, nb != 0);
  }
  return 0;
}

int orc_combine(void* dst, const void* src) {
  auto* d = (Handle*)dst; auto* s = (const Handle*)src;
  if (d->mode != s->mode) return 1;
  if (d->mode == 0) orc::combine(d->f, s->f); else orc::combine(d->d, s->d);
  return 0;
}

// Thread-local states over contiguous shards, merged in thread order — DuckDB's model
// (SURVEY.md §2 "Parallelism strategies").  Used for the CPU baseline.
// per_row_pointers != 0: through run_update_ptr (a per-chunk state-pointer vector read per row).
int orc_update_mt_ex(void* state, const float* const* num, int n, const int32_t* const* cat, int m,
                     int64_t rows, int nb, int threads, int per_row_pointers);
int orc_update_mt(void* state, const float* const* num, int n, const int32_t* const* cat, int m,
                  int64_t rows, int nb, int threads) {
  return orc_update_mt_ex(state, num, n, cat, m, rows, nb, threads, 0);
}
int orc_update_mt_ex(void* state, const float* const* num, int n, const int32_t* const* cat, int m,
                     int64_t rows, int nb, int threads, int per_row_pointers) {
  auto* h = (Handle*)state;
  if (threads < 1) threads = 1;
  // (volatile store: the optimiser must not prove the vector is all zeros)
  static std::vector<int32_t> chunk_sel;
  if (per_row_pointers && chunk_sel.empty()) {
    chunk_sel.assign(orc::CHUNK, 0);
    volatile int32_t* vs = chunk_sel.data();
    for (int64_t i = 0; i < orc::CHUNK; i++) vs[i] = 0;
  }
  const int32_t* sel = per_row_pointers ? chunk_sel.data() : nullptr;
  std::vector<Handle> local(threads);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; t++) {
    local[t].mode = h->mode;
    int64_t lo = rows * t / threads, hi = rows * (t + 1) / threads;
    pool.emplace_back([=, &local]() {
      if (h->mode == 0) {
        auto* s = &local[t].f;
        if (sel) orc::run_update_ptr<float, int32_t>(&s, sel, num, n, cat, m, lo, hi, nb != 0);
        else orc::run_update<float, int32_t>(&s, nullptr, num, n, cat, m, lo, hi, nb != 0);
      } else {
        auto* s = &local[t].d;
        if (sel) orc::run_update_ptr<double, int64_t>(&s, sel, num, n, cat, m, lo, hi, nb != 0);
        else orc::run_update<double, int64_t>(&s, nullptr, num, n, cat, m, lo, hi, nb != 0);
      }
    });
  }
  for (auto& th : pool) th.join();
  for (int t = 0; t < threads; t++) orc_combine(h, &local[t]);
  return 0;
}

static thread_local std::vector<double> g_scratch;

// Two-call protocol everywhere: out == nullptr -> returns the number of doubles needed.
int64_t orc_finalize(const void* state, double* out) {
  auto* h = (const Handle*)state;
  if (h->mode == 0) orc::finalize(h->f, g_scratch); else orc::finalize(h->d, g_scratch);
  if (out) std::memcpy(out, g_scratch.data(), g_scratch.size() * sizeof(double));
  return (int64_t)g_scratch.size();
}

// One blob per input row, concatenated; offsets[rows+1].
int64_t orc_lift(const float* const* num, int n, const int32_t* const* cat, int m, int64_t rows,
                 int nb, double* out, int64_t* offsets) {
  g_scratch.clear();
  orc::Triple t;
  for (int64_t i = 0; i < rows; i++) {
    if (offsets) offsets[i] = (int64_t)g_scratch.size();
    orc::lift_row(num, n, cat, m, i, nb != 0, t);
    orc::emit(t, g_scratch);
  }
  if (offsets) offsets[rows] = (int64_t)g_scratch.size();
  if (out) std::memcpy(out, g_scratch.data(), g_scratch.size() * sizeof(double));
  return (int64_t)g_scratch.size();
}

// sum_triple / sum_nb_agg update: add `count` blobs (concatenated) into the state.
int orc_sum_blobs(void* state, const double* blobs, const int64_t* offsets, int64_t count) {
  auto* h = (Handle*)state;
  orc::Triple t;
  for (int64_t i = 0; i < count; i++) {
    orc::parse(blobs + offsets[i], t);
    if (h->mode == 0) orc::sum_one(h->f, t); else orc::sum_one(h->d, t);
  }
  return 0;
}

int64_t orc_multiply(const double* a, const double* b, int mode, double* out) {
  orc::Triple A, B, R;
  orc::parse(a, A); orc::parse(b, B);
  if (mode == 0) orc::multiply<float>(A, B, R); else orc::multiply<double>(A, B, R);
  g_scratch.clear(); orc::emit(R, g_scratch);
  if (out) std::memcpy(out, g_scratch.data(), g_scratch.size() * sizeof(double));
  return (int64_t)g_scratch.size();
}

int64_t orc_add_sub(const double* a, const double* b, int sub, int mode, double* out) {
  orc::Triple A, B, R;
  orc::parse(a, A); orc::parse(b, B);
  if (mode == 0) orc::add_sub<float>(A, B, sub != 0, R); else orc::add_sub<double>(A, B, sub != 0, R);
  g_scratch.clear(); orc::emit(R, g_scratch);
  if (out) std::memcpy(out, g_scratch.data(), g_scratch.size() * sizeof(double));
  return (int64_t)g_scratch.size();
}

}  // extern "C"
