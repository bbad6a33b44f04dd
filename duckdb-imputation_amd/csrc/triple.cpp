// Host-side triple containers and scalar ring ops (see triple.hpp).
#include "triple.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

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
namespace {

void put_number(std::string &out, double v, bool as_float) {
  char buf[40];
  if (!as_float) { snprintf(buf, sizeof buf, "%lld", (long long)v); out += buf; return; }
  if (std::isnan(v)) { out += "nan"; return; }
  if (std::isinf(v)) { out += v > 0 ? "inf" : "-inf"; return; }
  for (int prec = 6; prec <= 17; prec++) {            // shortest form that reads back as the same double
    snprintf(buf, sizeof buf, "%.*g", prec, v);
    if (strtod(buf, nullptr) == v) break;
  }
  out += buf;
  if (!strpbrk(buf, ".eEn")) out += ".0";             // 15 -> 15.0, as DuckDB prints a FLOAT
}

struct TextCursor {
  const char *p, *end;
  std::string *err;
  void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
  bool eat(char c) { ws(); if (p < end && *p == c) { p++; return true; } return false; }
  bool need(char c) { if (eat(c)) return true; *err = std::string("triple text: expected '") + c + "'"; return false; }
  bool name(std::string &out) {
    ws();
    if (p >= end || (*p != '\'' && *p != '"')) { *err = "triple text: expected a quoted field name"; return false; }
    const char q = *p++;
    const char *s0 = p;
    while (p < end && *p != q) p++;
    if (p >= end) { *err = "triple text: unterminated field name"; return false; }
    out.assign(s0, p++);
    return true;
  }
  bool number(double &v) {
    ws();
    if (end - p >= 3 && !strncmp(p, "nan", 3)) { v = std::nan(""); p += 3; return true; }
    if (end - p >= 3 && !strncmp(p, "inf", 3)) { v = INFINITY; p += 3; return true; }
    if (end - p >= 4 && !strncmp(p, "-inf", 4)) { v = -INFINITY; p += 4; return true; }
    std::string tok;
    while (p < end && (isdigit((unsigned char)*p) || *p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E')) tok += *p++;
    if (tok.empty()) { *err = "triple text: expected a number"; return false; }
    char *stop = nullptr;
    v = strtod(tok.c_str(), &stop);
    if (!stop || *stop) { *err = "triple text: malformed number '" + tok + "'"; return false; }
    return true;
  }
  // [ number, ... ]
  bool numbers(std::vector<double> &out) {
    out.clear();
    if (!need('[')) return false;
    if (eat(']')) return true;
    do { double v; if (!number(v)) return false; out.push_back(v); } while (eat(','));
    return need(']');
  }
  // { 'a': number, 'b': number [, 'c': number] } with the given field names, in any order
  bool record(const char *const *names, int count, double *vals) {
    if (!need('{')) return false;
    bool seen[3] = {false, false, false};
    do {
      std::string nm;
      if (!name(nm) || !need(':')) return false;
      int idx = -1;
      for (int i = 0; i < count; i++) if (nm == names[i]) idx = i;
      if (idx < 0) { *err = "triple text: unexpected field '" + nm + "'"; return false; }
      if (!number(vals[idx])) return false;
      seen[idx] = true;
    } while (eat(','));
    for (int i = 0; i < count; i++) if (!seen[i]) { *err = std::string("triple text: missing field '") + names[i] + "'"; return false; }
    return need('}');
  }
};

}  // namespace

std::string triple_to_text(const ListTriple &t, bool aggregate_names) {
  std::string o = "{'N': ";
  put_number(o, t.N, false);
  auto dense = [&](const char *name, const std::vector<double> &v) {
    o += ", '"; o += name; o += "': [";
    for (size_t i = 0; i < v.size(); i++) { if (i) o += ", "; put_number(o, v[i], true); }
    o += "]";
  };
  dense(aggregate_names ? "lin_agg" : "lin_num", t.lin);
  dense(aggregate_names ? "quad_agg" : "quad_num", t.quad);
  auto kv = [&](const char *name, const std::vector<std::vector<KeyVal>> &lists) {
    o += ", '"; o += name; o += "': [";
    for (size_t l = 0; l < lists.size(); l++) {
      o += l ? ", [" : "[";
      for (size_t e = 0; e < lists[l].size(); e++) {
        o += e ? ", {'key': " : "{'key': ";
        put_number(o, lists[l][e].key, false);
        o += ", 'value': ";
        put_number(o, lists[l][e].val, true);
        o += "}";
      }
      o += "]";
    }
    o += "]";
  };
  kv("lin_cat", t.lin_cat);
  if (!t.kind) {
    kv("quad_num_cat", t.num_cat);
    o += ", 'quad_cat': [";
    for (size_t l = 0; l < t.cat_cat.size(); l++) {
      o += l ? ", [" : "[";
      for (size_t e = 0; e < t.cat_cat[l].size(); e++) {
        o += e ? ", {'key1': " : "{'key1': ";
        put_number(o, t.cat_cat[l][e].k1, false);
        o += ", 'key2': ";
        put_number(o, t.cat_cat[l][e].k2, false);
        o += ", 'value': ";
        put_number(o, t.cat_cat[l][e].val, true);
        o += "}";
      }
      o += "]";
    }
    o += "]";
  }
  o += "}";
  return o;
}

bool triple_from_text(const char *text, size_t len, ListTriple &t, std::string &err) {
  TextCursor c{text, text + len, &err};
  t = ListTriple();
  bool have_N = false, have_lin = false, have_quad = false, have_lc = false, have_nc = false, have_cc = false;
  if (!c.need('{')) return false;
  do {
    std::string nm;
    if (!c.name(nm) || !c.need(':')) return false;
    if (nm == "N") { if (!c.number(t.N)) return false; have_N = true; }
    else if (nm == "lin_agg" || nm == "lin_num") { if (!c.numbers(t.lin)) return false; have_lin = true; }
    else if (nm == "quad_agg" || nm == "quad_num") { if (!c.numbers(t.quad)) return false; have_quad = true; }
    else if (nm == "lin_cat" || nm == "quad_num_cat") {
      auto &dst = nm == "lin_cat" ? t.lin_cat : t.num_cat;
      (nm == "lin_cat" ? have_lc : have_nc) = true;
      if (!c.need('[')) return false;
      if (!c.eat(']')) {
        do {
          dst.emplace_back();
          if (!c.need('[')) return false;
          if (c.eat(']')) continue;
          do {
            static const char *const names[] = {"key", "value"};
            double v[2];
            if (!c.record(names, 2, v)) return false;
            dst.back().push_back({(int32_t)v[0], v[1]});
          } while (c.eat(','));
          if (!c.need(']')) return false;
        } while (c.eat(','));
        if (!c.need(']')) return false;
      }
    } else if (nm == "quad_cat") {
      have_cc = true;
      if (!c.need('[')) return false;
      if (!c.eat(']')) {
        do {
          t.cat_cat.emplace_back();
          if (!c.need('[')) return false;
          if (c.eat(']')) continue;
          do {
            static const char *const names[] = {"key1", "key2", "value"};
            double v[3];
            if (!c.record(names, 3, v)) return false;
            t.cat_cat.back().push_back({(int32_t)v[0], (int32_t)v[1], v[2]});
          } while (c.eat(','));
          if (!c.need(']')) return false;
        } while (c.eat(','));
        if (!c.need(']')) return false;
      }
    } else { err = "triple text: unknown field '" + nm + "'"; return false; }
  } while (c.eat(','));
  if (!c.need('}')) return false;
  c.ws();
  if (c.p != c.end) { err = "triple text: trailing characters"; return false; }
  if (!have_N || !have_lin || !have_quad || !have_lc) { err = "triple text: N, lin, quad and lin_cat are required"; return false; }
  if (have_nc != have_cc) { err = "triple text: quad_num_cat and quad_cat come together"; return false; }
  t.kind = have_cc ? 0 : 1;
  t.n = (int)t.lin.size();
  t.m = (int)t.lin_cat.size();
  const size_t want_quad = t.kind ? (size_t)t.n : tri(t.n);
  if (t.quad.size() != want_quad) { err = "triple text: quad list length does not fit the lin list"; return false; }
  if (!t.kind && (t.num_cat.size() != (size_t)t.n * t.m || t.cat_cat.size() != tri(t.m))) {
    err = "triple text: quad_num_cat / quad_cat list counts do not fit n and m";
    return false;
  }
  return true;
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
