// Host-side triple containers and scalar ring ops (see triple.hpp).
#include "triple.hpp"

#include <algorithm>
#include <cmath>

namespace cofactor {

namespace {
struct Reader {               // cursor over a blob whose extent blob_len has validated
  const double *p;
  double next() { return *p++; }
};

bool small_nonneg_int(double v, double hi) { return v >= 0 && v <= hi && v == std::floor(v); }
}  // namespace

// Walks the self-describing blob without ever reading at or beyond b[cap]: every list header is
// checked against what is left.  0 = malformed or truncated.
uint64_t blob_len(const double *b, uint64_t cap) {
  if (!b || cap < 4) return 0;
  if (!(b[0] == 0 || b[0] == 1) || !small_nonneg_int(b[1], 1 << 20) ||
      !small_nonneg_int(b[2], 1 << 20))
    return 0;
  const int kind = (int)b[0];
  const uint64_t n = (uint64_t)b[1], m = (uint64_t)b[2];
  uint64_t pos = 4 + n + (kind ? n : tri(n));
  if (pos > cap) return 0;
  auto skip = [&](uint64_t lists, uint64_t width) {
    for (uint64_t l = 0; l < lists; l++) {
      if (pos >= cap) return false;
      const double len = b[pos];
      if (!small_nonneg_int(len, 4e15)) return false;
      const uint64_t body = (uint64_t)len * width;
      if (body > cap - pos - 1) return false;
      pos += 1 + body;
    }
    return true;
  };
  if (!skip(m, 2)) return 0;
  if (kind == 0) {
    if (!skip(n * m, 2)) return 0;
    if (!skip(tri(m), 3)) return 0;
  }
  return pos;
}

bool blob_decode(const double *b, uint64_t cap, ListTriple &t, std::string &err) {
  if (blob_len(b, cap) == 0) { err = "malformed or truncated triple blob"; return false; }
  Reader r{b};
  t.kind = (int)r.next(); t.n = (int)r.next(); t.m = (int)r.next(); t.N = r.next();
  t.lin.resize(t.n);
  for (auto &v : t.lin) v = r.next();
  t.quad.resize(t.kind ? t.n : tri(t.n));
  for (auto &v : t.quad) v = r.next();
  auto kv = [&](std::vector<std::vector<KeyVal>> &dst, size_t lists) {
    dst.assign(lists, {});
    for (auto &lst : dst) {
      lst.resize((size_t)r.next());
      for (auto &e : lst) { e.key = (int32_t)r.next(); e.val = r.next(); }
    }
  };
  kv(t.lin_cat, t.m);
  t.num_cat.clear(); t.cat_cat.clear();
  if (t.kind == 0) {
    kv(t.num_cat, (size_t)t.n * t.m);
    t.cat_cat.assign(tri(t.m), {});
    for (auto &lst : t.cat_cat) {
      lst.resize((size_t)r.next());
      for (auto &e : lst) { e.k1 = (int32_t)r.next(); e.k2 = (int32_t)r.next(); e.val = r.next(); }
    }
  }
  return true;
}

void blob_encode(const ListTriple &t, std::vector<double> &out) {
  out.push_back(t.kind); out.push_back(t.n); out.push_back(t.m); out.push_back(t.N);
  out.insert(out.end(), t.lin.begin(), t.lin.end());
  out.insert(out.end(), t.quad.begin(), t.quad.end());
  auto kv = [&](const std::vector<std::vector<KeyVal>> &src) {
    for (auto const &lst : src) {
      out.push_back((double)lst.size());
      for (auto const &e : lst) { out.push_back(e.key); out.push_back(e.val); }
    }
  };
  kv(t.lin_cat);
  if (t.kind) return;
  kv(t.num_cat);
  for (auto const &lst : t.cat_cat) {
    out.push_back((double)lst.size());
    for (auto const &e : lst) { out.push_back(e.k1); out.push_back(e.k2); out.push_back(e.val); }
  }
}

// ---------------------------------------------------------------------------------------------
void HostTriple::shape(int kind_, int n_, int m_) {
  kind = kind_; n = n_; m = m_;
  clear();
}

void HostTriple::clear() {
  N = 0;
  lin.assign(n, 0.0);
  quad.assign(kind ? n : tri(n), 0.0);
  col.assign(m, {});
  pair.assign(kind ? 0 : tri(m), {});
}

bool HostTriple::add_list(const ListTriple &t, std::string &err) {
  if (t.kind != kind || t.n != n || t.m != m) {
    err = "triple shape mismatch: state is (" + std::to_string(n) + "," + std::to_string(m) +
          ") kind " + std::to_string(kind) + ", input is (" + std::to_string(t.n) + "," +
          std::to_string(t.m) + ") kind " + std::to_string(t.kind);
    return false;
  }
  N += t.N;
  for (int k = 0; k < n; k++) lin[k] += t.lin[k];
  for (size_t k = 0; k < quad.size(); k++) quad[k] += t.quad[k];
  const size_t width = kind ? 1 : (size_t)n + 1;
  for (int c = 0; c < m; c++) {
    auto const &keys = t.lin_cat[c];
    for (size_t e = 0; e < keys.size(); e++) {
      auto &slot = col[c][keys[e].key];
      if (slot.empty()) slot.assign(width, 0.0);
      slot[0] += keys[e].val;
      if (!kind)
        for (int k = 0; k < n; k++) {
          auto const &lst = t.num_cat[(size_t)k * m + c];
          // the reference asserts the same alignment (sum.cpp:212-213)
          if (lst.size() != keys.size() || lst[e].key != keys[e].key) {
            err = "quad_num_cat list not key-aligned with lin_cat";
            return false;
          }
          slot[k + 1] += lst[e].val;
        }
    }
  }
  if (!kind)
    for (size_t q = 0; q < pair.size(); q++)
      for (auto const &e : t.cat_cat[q]) pair[q][{e.k1, e.k2}] += e.val;
  return true;
}

bool HostTriple::add(const HostTriple &o, std::string &err) {
  if (o.kind != kind || o.n != n || o.m != m) { err = "combine: state shape mismatch"; return false; }
  N += o.N;
  for (int k = 0; k < n; k++) lin[k] += o.lin[k];
  for (size_t k = 0; k < quad.size(); k++) quad[k] += o.quad[k];
  for (int c = 0; c < m; c++)
    for (auto const &kv : o.col[c]) {
      auto &slot = col[c][kv.first];
      if (slot.empty()) slot = kv.second;
      else for (size_t k = 0; k < slot.size(); k++) slot[k] += kv.second[k];
    }
  for (size_t q = 0; q < pair.size(); q++)
    for (auto const &kv : o.pair[q]) pair[q][kv.first] += kv.second;
  return true;
}

void HostTriple::encode(std::vector<double> &out) const {
  encode_without_pairs(out);
  if (kind) return;
  for (auto const &tab : pair) {
    out.push_back((double)tab.size());
    for (auto const &kv : tab) {
      out.push_back(kv.first.first); out.push_back(kv.first.second); out.push_back(kv.second);
    }
  }
}

void HostTriple::encode_without_pairs(std::vector<double> &out) const {
  out.push_back(kind); out.push_back(n); out.push_back(m); out.push_back(N);
  out.insert(out.end(), lin.begin(), lin.end());
  out.insert(out.end(), quad.begin(), quad.end());
  for (int c = 0; c < m; c++) {
    out.push_back((double)col[c].size());
    for (auto const &kv : col[c]) { out.push_back(kv.first); out.push_back(kv.second[0]); }
  }
  if (kind) return;
  for (int k = 0; k < n; k++)
    for (int c = 0; c < m; c++) {
      out.push_back((double)col[c].size());
      for (auto const &kv : col[c]) { out.push_back(kv.first); out.push_back(kv.second[k + 1]); }
    }
}

// ---------------------------------------------------------------------------------------------
void lift_row(const float *const *num, int n, const int32_t *const *cat, int m, uint64_t row,
              int kind, ListTriple &t) {
  t.kind = kind; t.n = n; t.m = m; t.N = 1;
  t.lin.resize(n);
  for (int k = 0; k < n; k++) t.lin[k] = num[k][row];
  t.quad.clear();
  t.lin_cat.assign(m, {});
  for (int c = 0; c < m; c++) t.lin_cat[c].push_back({cat[c][row], 1.0});
  t.num_cat.clear(); t.cat_cat.clear();
  if (kind) {
    for (int k = 0; k < n; k++) t.quad.push_back((double)(num[k][row] * num[k][row]));
    return;
  }
  // products are formed in float, as the reference's FLOAT result column holds them
  for (int j = 0; j < n; j++)
    for (int k = j; k < n; k++) t.quad.push_back((double)(num[j][row] * num[k][row]));
  t.num_cat.assign((size_t)n * m, {});
  for (int j = 0; j < n; j++)
    for (int c = 0; c < m; c++) t.num_cat[(size_t)j * m + c].push_back({cat[c][row], (double)num[j][row]});
  t.cat_cat.assign(tri(m), {});
  size_t q = 0;
  for (int c1 = 0; c1 < m; c1++)
    for (int c2 = c1; c2 < m; c2++, q++) t.cat_cat[q].push_back({cat[c1][row], cat[c2][row], 1.0});
}

static std::vector<KeyVal> scale_list(const std::vector<KeyVal> &src, double f) {
  std::vector<KeyVal> dst(src);
  for (auto &e : dst) e.val *= f;
  return dst;
}

bool multiply(const ListTriple &A, const ListTriple &B, ListTriple &R, std::string &err) {
  if (A.kind != B.kind) { err = "multiply: triple kinds differ"; return false; }
  R = ListTriple();
  R.kind = A.kind; R.n = A.n + B.n; R.m = A.m + B.m;
  R.N = A.N * B.N;
  for (double v : A.lin) R.lin.push_back(v * B.N);
  for (double v : B.lin) R.lin.push_back(v * A.N);
  for (auto const &l : A.lin_cat) R.lin_cat.push_back(scale_list(l, B.N));
  for (auto const &l : B.lin_cat) R.lin_cat.push_back(scale_list(l, A.N));
  if (A.kind) {
    for (double v : A.quad) R.quad.push_back(v * B.N);
    for (double v : B.quad) R.quad.push_back(v * A.N);
    return true;
  }
  // upper triangle of [[N_B Q_A, lin_A (x) lin_B], [., N_A Q_B]], row-major
  size_t q = 0;
  for (int j = 0; j < A.n; j++) {
    for (int k = j; k < A.n; k++) R.quad.push_back(A.quad[q++] * B.N);
    for (int k = 0; k < B.n; k++) R.quad.push_back(A.lin[j] * B.lin[k]);
  }
  for (double v : B.quad) R.quad.push_back(v * A.N);
  // numeric-major over (A|B), categorical-minor over (A|B)
  for (int j = 0; j < A.n; j++) {
    for (int c = 0; c < A.m; c++) R.num_cat.push_back(scale_list(A.num_cat[(size_t)j * A.m + c], B.N));
    for (int c = 0; c < B.m; c++) R.num_cat.push_back(scale_list(B.lin_cat[c], A.lin[j]));
  }
  for (int j = 0; j < B.n; j++) {
    for (int c = 0; c < A.m; c++) R.num_cat.push_back(scale_list(A.lin_cat[c], B.lin[j]));
    for (int c = 0; c < B.m; c++) R.num_cat.push_back(scale_list(B.num_cat[(size_t)j * B.m + c], A.N));
  }
  // upper triangle over the joined categorical columns
  q = 0;
  for (int c1 = 0; c1 < A.m; c1++) {
    for (int c2 = c1; c2 < A.m; c2++, q++) {
      std::vector<PairVal> l(A.cat_cat[q]);
      for (auto &e : l) e.val *= B.N;
      R.cat_cat.push_back(std::move(l));
    }
    for (int c2 = 0; c2 < B.m; c2++) {
      std::vector<PairVal> l;
      for (auto const &ka : A.lin_cat[c1])
        for (auto const &kb : B.lin_cat[c2]) l.push_back({ka.key, kb.key, ka.val * kb.val});
      R.cat_cat.push_back(std::move(l));
    }
  }
  for (auto const &src : B.cat_cat) {
    std::vector<PairVal> l(src);
    for (auto &e : l) e.val *= A.N;
    R.cat_cat.push_back(std::move(l));
  }
  return true;
}

bool add_sub(const ListTriple &A, const ListTriple &B, bool subtract, ListTriple &R,
             std::string &warn) {
  const double sgn = subtract ? -1.0 : 1.0;
  R = ListTriple();
  R.kind = A.kind; R.n = std::max(A.n, B.n); R.m = std::max(A.m, B.m);
  R.N = A.N + sgn * B.N;
  auto dense = [&](const std::vector<double> &a, const std::vector<double> &b, std::vector<double> &r) {
    if (!a.empty() && !b.empty()) {
      r.resize(a.size());
      for (size_t i = 0; i < a.size(); i++) r[i] = a[i] + sgn * (i < b.size() ? b[i] : 0.0);
    } else {
      r = a.empty() ? b : a;   // an empty side is copied from the other (sum.cpp:82-93)
    }
  };
  dense(A.lin, B.lin, R.lin);
  dense(A.quad, B.quad, R.quad);
  auto kv = [&](const std::vector<std::vector<KeyVal>> &a, const std::vector<std::vector<KeyVal>> &b,
                std::vector<std::vector<KeyVal>> &r) {
    if (a.empty() || b.empty()) { r = a.empty() ? b : a; return; }
    for (size_t l = 0; l < a.size(); l++) {
      std::map<int32_t, double> acc;
      for (auto const &e : a[l]) acc[e.key] = e.val;
      if (l < b.size())
        for (auto const &e : b[l]) {
          auto it = acc.find(e.key);
          if (it != acc.end()) it->second += sgn * e.val;
          else if (!subtract) acc[e.key] = e.val;
          else warn = "subtract: key " + std::to_string(e.key) + " is not present in first triple";
        }
      r.emplace_back();
      for (auto const &e : acc) r.back().push_back({e.first, e.second});
    }
  };
  kv(A.lin_cat, B.lin_cat, R.lin_cat);
  if (A.kind) return true;
  kv(A.num_cat, B.num_cat, R.num_cat);
  if (A.cat_cat.empty() || B.cat_cat.empty()) { R.cat_cat = A.cat_cat.empty() ? B.cat_cat : A.cat_cat; return true; }
  for (size_t l = 0; l < A.cat_cat.size(); l++) {
    std::map<std::pair<int32_t, int32_t>, double> acc;
    for (auto const &e : A.cat_cat[l]) acc[{e.k1, e.k2}] = e.val;
    if (l < B.cat_cat.size())
      for (auto const &e : B.cat_cat[l]) {
        auto it = acc.find({e.k1, e.k2});
        if (it != acc.end()) it->second += sgn * e.val;
        else if (!subtract) acc[{e.k1, e.k2}] = e.val;
        else warn = "subtract: key pair (" + std::to_string(e.k1) + "," + std::to_string(e.k2) +
                    ") is not present in first triple";
      }
    R.cat_cat.emplace_back();
    for (auto const &e : acc) R.cat_cat.back().push_back({e.first.first, e.first.second, e.second});
  }
  return true;
}

}  // namespace cofactor
