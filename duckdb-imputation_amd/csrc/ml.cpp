// Triple consumers on the host (see ml.hpp).  Everything is fp64 over a p x p matrix with
// p = 1 + n + #distinct keys, i.e. independent of the row count: the heavy lifting was the
// aggregate.
#include "ml.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>

namespace cofactor {

void onehot_layout(const ListTriple &t, OneHot &oh) {
  oh.begin.assign(1, 0);
  oh.keys.clear();
  for (int c = 0; c < t.m; c++) {
    std::vector<int64_t> k;
    for (auto const &e : t.lin_cat[c]) k.push_back(e.key);
    std::sort(k.begin(), k.end());
    k.erase(std::unique(k.begin(), k.end()), k.end());
    oh.keys.insert(oh.keys.end(), k.begin(), k.end());
    oh.begin.push_back((uint32_t)oh.keys.size());
  }
}

namespace {

// Column of (key column c, key) in a sigma matrix that leaves key column `skip` out; -1 when the
// key is not in the layout.
struct Slots {
  const OneHot &oh;
  int n, skip;
  size_t skipped;  // keys of the skipped column
  Slots(const OneHot &o, int n_, int skip_) : oh(o), n(n_), skip(skip_), skipped(0) {
    if (skip >= 0) skipped = oh.begin[skip + 1] - oh.begin[skip];
  }
  long at(int c, int64_t key) const {
    auto b = oh.keys.begin() + oh.begin[c], e = oh.keys.begin() + oh.begin[c + 1];
    auto it = std::lower_bound(b, e, key);
    if (it == e || *it != key) return -1;
    size_t pos = 1 + (size_t)n + (size_t)(it - oh.keys.begin());
    if (skip >= 0 && c > skip) pos -= skipped;
    return (long)pos;
  }
  size_t width() const { return 1 + (size_t)n + oh.keys.size() - skipped; }
};

}  // namespace

size_t build_sigma(const ListTriple &t, const OneHot &oh, int skip_cat, std::vector<double> &sigma) {
  const Slots slot(oh, t.n, skip_cat);
  const size_t p = slot.width(), n = (size_t)t.n, m = (size_t)t.m;
  sigma.assign(p * p, 0.0);
  sigma[0] = t.N;
  for (size_t j = 0; j < n; j++) sigma[1 + j] = sigma[(1 + j) * p] = t.lin[j];
  size_t q = 0;
  for (size_t j = 0; j < n; j++)
    for (size_t k = j; k < n; k++, q++)
      sigma[(1 + j) * p + 1 + k] = sigma[(1 + k) * p + 1 + j] = t.quad[q];
  for (size_t c = 0; c < m; c++) {
    if ((int)c == skip_cat) continue;
    for (auto const &e : t.lin_cat[c]) {
      const long k = slot.at((int)c, e.key);
      if (k < 0) continue;
      sigma[k] = sigma[k * p] = sigma[k * p + k] = e.val;
    }
    for (size_t j = 0; j < n; j++)
      for (auto const &e : t.num_cat[j * m + c]) {
        const long k = slot.at((int)c, e.key);
        if (k < 0) continue;
        sigma[k * p + 1 + j] = sigma[(1 + j) * p + k] = e.val;
      }
  }
  q = 0;
  for (size_t c1 = 0; c1 < m; c1++)
    for (size_t c2 = c1; c2 < m; c2++, q++) {
      if ((int)c1 == skip_cat || (int)c2 == skip_cat) continue;
      for (auto const &e : t.cat_cat[q]) {
        const long a = slot.at((int)c1, e.k1), b = slot.at((int)c2, e.k2);
        if (a < 0 || b < 0) continue;
        sigma[a * p + b] = sigma[b * p + a] = e.val;
      }
    }
  return p;
}

namespace {

// standardize_sigma (ML/utils.cpp:580-597): the covariance-like matrix of the standardised
// columns, intercept row/column zeroed except [0][0].
void standardize(std::vector<double> &s, size_t p, std::vector<double> &mean,
                 std::vector<double> &sd) {
  mean.resize(p); sd.resize(p);
  for (size_t i = 0; i < p; i++) mean[i] = s[i] / s[0];
  for (size_t i = 0; i < p; i++) sd[i] = std::sqrt(s[i * p + i] / s[0] - mean[i] * mean[i]);
  for (size_t i = 1; i < p; i++)
    for (size_t j = 1; j < p; j++)
      s[i * p + j] = (s[i * p + j] - mean[i] * s[j] - mean[j] * s[i] + s[0] * mean[j] * mean[i]) /
                     (sd[i] * sd[j]);
  for (size_t i = 1; i < p; i++) s[i] = s[i * p] = 0;
}

bool cholesky_solve(const std::vector<double> &A, size_t p, std::vector<double> &B, size_t nrhs) {
  std::vector<double> L(A);
  double dmax = 0;
  for (size_t i = 0; i < p; i++) dmax = std::max(dmax, std::fabs(A[i * p + i]));
  const double floor_ = dmax * (double)p * DBL_EPSILON * 1e4;
  for (size_t j = 0; j < p; j++) {
    double d = L[j * p + j];
    for (size_t k = 0; k < j; k++) d -= L[j * p + k] * L[j * p + k];
    if (!(d > floor_)) return false;
    d = std::sqrt(d);
    L[j * p + j] = d;
    for (size_t i = j + 1; i < p; i++) {
      double v = L[i * p + j];
      for (size_t k = 0; k < j; k++) v -= L[i * p + k] * L[j * p + k];
      L[i * p + j] = v / d;
    }
  }
  for (size_t r = 0; r < nrhs; r++) {
    double *b = &B[r * p];
    for (size_t i = 0; i < p; i++) {
      double v = b[i];
      for (size_t k = 0; k < i; k++) v -= L[i * p + k] * b[k];
      b[i] = v / L[i * p + i];
    }
    for (size_t i = p; i-- > 0;) {
      double v = b[i];
      for (size_t k = i + 1; k < p; k++) v -= L[k * p + i] * b[k];
      b[i] = v / L[i * p + i];
    }
  }
  return true;
}

// Cyclic Jacobi: A = V diag(w) V^T, A destroyed, V row-major with eigenvectors in columns.
void jacobi_eigen(std::vector<double> &A, size_t p, std::vector<double> &V, std::vector<double> &w) {
  V.assign(p * p, 0.0);
  for (size_t i = 0; i < p; i++) V[i * p + i] = 1;
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = 0, diag = 0;
    for (size_t i = 0; i < p; i++) {
      diag += A[i * p + i] * A[i * p + i];
      for (size_t j = i + 1; j < p; j++) off += A[i * p + j] * A[i * p + j];
    }
    if (off <= diag * 1e-32 || off == 0) break;
    for (size_t q = 0; q + 1 < p; q++)
      for (size_t r = q + 1; r < p; r++) {
        const double aqr = A[q * p + r];
        if (aqr == 0) continue;
        const double theta = (A[r * p + r] - A[q * p + q]) / (2 * aqr);
        const double tt = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
        const double c = 1 / std::sqrt(tt * tt + 1), s = tt * c;
        for (size_t k = 0; k < p; k++) {  // columns q, r
          const double akq = A[k * p + q], akr = A[k * p + r];
          A[k * p + q] = c * akq - s * akr;
          A[k * p + r] = s * akq + c * akr;
        }
        for (size_t k = 0; k < p; k++) {  // rows q, r
          const double aqk = A[q * p + k], ark = A[r * p + k];
          A[q * p + k] = c * aqk - s * ark;
          A[r * p + k] = s * aqk + c * ark;
        }
        for (size_t k = 0; k < p; k++) {
          const double vkq = V[k * p + q], vkr = V[k * p + r];
          V[k * p + q] = c * vkq - s * vkr;
          V[k * p + r] = s * vkq + c * vkr;
        }
      }
  }
  w.resize(p);
  for (size_t i = 0; i < p; i++) w[i] = A[i * p + i];
}

}  // namespace

void solve_symmetric_min_norm(std::vector<double> &A, size_t p, std::vector<double> &B, size_t nrhs) {
  if (p == 0) return;
  if (cholesky_solve(A, p, B, nrhs)) return;
  std::vector<double> V, w;
  jacobi_eigen(A, p, V, w);
  double wmax = 0;
  for (double v : w) wmax = std::max(wmax, std::fabs(v));
  const double cut = wmax * DBL_EPSILON * (double)p;
  std::vector<double> y(p);
  for (size_t r = 0; r < nrhs; r++) {
    double *b = &B[r * p];
    for (size_t e = 0; e < p; e++) {
      double v = 0;
      for (size_t k = 0; k < p; k++) v += V[k * p + e] * b[k];
      y[e] = std::fabs(w[e]) > cut ? v / w[e] : 0.0;
    }
    for (size_t k = 0; k < p; k++) {
      double v = 0;
      for (size_t e = 0; e < p; e++) v += V[k * p + e] * y[e];
      b[k] = v;
    }
  }
}

namespace {

// v = Sigma * theta: the one product both the gradient and the objective need (the reference
// forms it twice per iteration, compute_gradient regression.cpp:27-46 and compute_error :48-77).
// Sigma is symmetric (build_sigma writes both halves), so v is also the sum of theta_j times ROW j:
// 64 entries of v stay in registers while the rows stream by — vertical vector operations only, no
// horizontal sum per row (the row-times-vector form with eight partial sums took 3.3 us at p = 171
// on the bench host, this one 1.2).  Compiled for AVX-512 / AVX2 as well and picked at load time
// (the descent calls this several hundred times on a p x p matrix: it IS the training time).  No
// contraction into FMAs: every clone rounds the same way, each v[i] is the sum over j in order.
#if defined(__x86_64__) && defined(__linux__)
__attribute__((target_clones("avx512f", "avx2", "default")))
#endif
void sigma_times_raw(size_t p, const double *__restrict__ s, const double *__restrict__ th, double *__restrict__ v) {
#pragma clang fp contract(off)
  constexpr size_t B = 64;
  for (size_t i0 = 0; i0 < p; i0 += B) {
    const size_t w = p - i0 < B ? p - i0 : B;
    double a[B];
    for (size_t u = 0; u < B; u++) a[u] = 0;
    if (w == B) {
      for (size_t j = 0; j < p; j++) {
        const double t = th[j];
        const double *row = s + j * p + i0;
        for (size_t u = 0; u < B; u++) a[u] += row[u] * t;
      }
    } else {
      for (size_t j = 0; j < p; j++) {
        const double t = th[j];
        const double *row = s + j * p + i0;
        for (size_t u = 0; u < w; u++) a[u] += row[u] * t;
      }
    }
    for (size_t u = 0; u < w; u++) v[i0 + u] = a[u];
  }
}
void sigma_times(size_t p, const std::vector<double> &s, const std::vector<double> &th,
                 std::vector<double> &v) {

  sigma_times_raw(p, s.data(), th.data(), v.data());
}

// 1/N * Sigma * theta with the label's entry forced to 0
void gradient(size_t p, size_t label, double N, const std::vector<double> &v, std::vector<double> &g) {
  if (N == 0.0) return;
  for (size_t i = 0; i < p; i++) g[i] = v[i] / N;
  g[label] = 0;
}

// (theta^T Sigma theta / N + lambda (|theta_1..|^2 - 1)) / 2
double objective(size_t p, double N, const std::vector<double> &v, const std::vector<double> &th,
                 double lambda) {
  if (N == 0.0) return 0;
  double e = 0;
  for (size_t i = 0; i < p; i++) e += th[i] * v[i];
  e /= N;
  double nrm = 0;
  for (size_t i = 1; i < p; i++) nrm += th[i] * th[i];
  nrm -= 1;
  return (e + lambda * nrm) / 2;
}

// Barzilai-Borwein step (compute_step_size, regression.cpp:79-105)
double bb_step(double step, size_t p, const std::vector<double> &th, const std::vector<double> &pth,
               const std::vector<double> &g, const std::vector<double> &pg) {
  double dss = 0, gss = 0, dgs = 0;
  for (size_t i = 0; i < p; i++) {
    const double dp = th[i] - pth[i], dg = g[i] - pg[i];
    dss += dp * dp; gss += dg * dg; dgs += dp * dg;
  }
  if (dgs == 0.0 || gss == 0.0) return step;
  const double ts = dss / dgs, tm = dgs / gss;
  if (tm < 0.0 || ts < 0.0) return step;
  return (tm / ts > 0.5) ? tm : ts - 0.5 * tm;
}

}  // namespace

bool linreg_train(const ListTriple &t, int label0, float step_size, float lambda,
                  int max_iterations, bool compute_variance, bool normalize,
                  std::vector<float> &out, std::string &err) {
  if (t.kind != 0) { err = "linreg_train needs a full triple, not an nb aggregate"; return false; }
  if (label0 < 0 || label0 >= t.n) { err = "label is not a numeric column of the triple"; return false; }
  OneHot oh;
  onehot_layout(t, oh);
  std::vector<double> sigma;
  const size_t p = build_sigma(t, oh, -1, sigma);
  std::vector<double> mean, sd;
  if (normalize) standardize(sigma, p, mean, sd);

  std::vector<double> g(p, 0), pg(p, 0), th(p, 0), pth(p, 0), upd(p, 0), sv(p, 0);
  const size_t label = (size_t)label0 + 1;  // slot 0 is the intercept
  const double N = sigma[0];
  th[label] = pth[label] = -1;
  sigma_times(p, sigma, th, sv);
  gradient(p, label, N, sv, g);
  double gnorm = g[0] * g[0];
  for (size_t i = 1; i < p; i++) {
    const double u = g[i] + lambda * th[i];
    gnorm += u * u;
  }
  gnorm -= (double)lambda * lambda;
  const double first_gnorm = std::sqrt(gnorm);
  double prev_err = objective(p, N, sv, th, lambda);

  // the reference keeps the step in a float (regression.cpp:115); so do we, the trajectory of the
  // descent depends on it
  float step = step_size;
  int it = 1, stalled = 0;
  do {
    upd[0] = g[0];
    gnorm = upd[0] * upd[0];
    pth[0] = th[0]; pg[0] = g[0];
    th[0] -= step * upd[0];
    double dnorm = upd[0] * upd[0];
    for (size_t i = 1; i < p; i++) {
      upd[i] = g[i] + lambda * th[i];
      gnorm += upd[i] * upd[i];
      pth[i] = th[i]; pg[i] = g[i];
      th[i] -= step * upd[i];
      dnorm += upd[i] * upd[i];
    }
    th[label] = -1;
    gnorm -= (double)lambda * lambda;
    dnorm = step * std::sqrt(dnorm);
    sigma_times(p, sigma, th, sv);
    double e = objective(p, N, sv, th, lambda);
    int back = 0;
    while (e > prev_err - (step / 2) * gnorm && back < 500) {  // backtracking line search
      step /= 2;
      dnorm = 0;
      for (size_t i = 0; i < p; i++) {
        const double np = pth[i] - step * upd[i], dp = th[i] - np;
        th[i] = np;
        dnorm += dp * dp;
      }
      dnorm = std::sqrt(dnorm);
      th[label] = -1;
      sigma_times(p, sigma, th, sv);
      e = objective(p, N, sv, th, lambda);
      back++;
    }
    gnorm = std::sqrt(gnorm);
    if (dnorm < 1e-20 || gnorm / (first_gnorm + 0.001) < 1e-8) break;
    // The reference's two criteria alone can idle until max_iterations: the objective reaches its
    // last bit with the gradient ratio a hair above 1e-8 and steps of 1e-19 (seen on uniform
    // data: 31 useful iterations, 9 969 idle ones).  Three iterations in a row without any
    // decrease of the objective end the descent; the parameters no longer move by then.
    stalled = (e < prev_err) ? 0 : stalled + 1;
    if (stalled >= 3) break;
    gradient(p, label, N, sv, g);   // sv is Sigma * theta of the accepted point
    step = (float)bb_step(step, p, th, pth, g, pg);
    prev_err = e;
    it++;
  } while (it < max_iterations);

  double variance = 0;
  if (compute_variance) {  // theta^T Sigma theta / N with theta[label] = -1: the residual variance
    th[label] = -1;
    sigma_times(p, sigma, th, sv);
    for (size_t i = 0; i < p; i++) variance += th[i] * sv[i];
    variance /= t.N;
  }
  if (normalize) {
    for (size_t i = 1; i < p; i++) th[i] = th[i] / sd[i] * sd[label];
    th[0] = th[0] * sd[label] + mean[label];
  }

  // [m, begin[0..m], keys, coefficients without the label's, (means without intercept and
  // label), (sqrt(variance))]   (regression.cpp:313-353)
  out.clear();
  out.push_back((float)t.m);
  if (t.m > 0) {
    for (uint32_t b : oh.begin) out.push_back((float)b);
    for (int64_t k : oh.keys) out.push_back((float)k);
  }
  for (size_t i = 0; i < p; i++)
    if (i != label) out.push_back((float)th[i]);
  if (normalize)
    for (size_t i = 1; i < p; i++)
      if (i != label) out.push_back((float)mean[i]);
  if (compute_variance) out.push_back((float)std::sqrt(variance));
  return true;
}

bool lda_train(const ListTriple &t, int label, float shrinkage, bool normalize,
               std::vector<float> &out, std::string &err) {
  if (t.kind != 0) { err = "lda_train needs a full triple, not an nb aggregate"; return false; }
  if (label < 0 || label >= t.m) { err = "label is not a key column of the triple"; return false; }
  OneHot oh;
  onehot_layout(t, oh);
  std::vector<double> sigma;
  const size_t p1 = build_sigma(t, oh, label, sigma);
  const size_t n = (size_t)t.n, m = (size_t)t.m;
  const size_t lb = oh.begin[label], le = oh.begin[label + 1], C = le - lb;
  if (C == 0 || p1 < 2) { err = "lda_train: the label column has no keys or there are no features"; return false; }
  const Slots slot(oh, t.n, label);
  auto cls = [&](int64_t key) -> long {
    auto b = oh.keys.begin() + lb, e = oh.keys.begin() + le;
    auto it = std::lower_bound(b, e, key);
    return (it == e || *it != key) ? -1 : (long)(it - b);
  };

  // per class: [count, sum of every feature column]   (build_sum_vector, ML/lda.cpp:50-152; the
  // key-column slots use sigma's layout, which is what the reference's arithmetic downstream
  // assumes — its own indexing only agrees with that when the label is the last key column)
  std::vector<double> sv(C * p1, 0.0);
  for (auto const &e : t.lin_cat[label]) sv[(size_t)cls(e.key) * p1] = e.val;
  for (size_t j = 0; j < n; j++)
    for (auto const &e : t.num_cat[j * m + label]) sv[(size_t)cls(e.key) * p1 + 1 + j] = e.val;
  size_t q = 0;
  for (size_t c1 = 0; c1 < m; c1++)
    for (size_t c2 = c1; c2 < m; c2++, q++) {
      if (c1 == c2 || ((int)c1 != label && (int)c2 != label)) continue;
      for (auto const &e : t.cat_cat[q]) {
        const bool first = (int)c1 == label;
        const long g = cls(first ? e.k1 : e.k2);
        const long k = slot.at((int)(first ? c2 : c1), first ? e.k2 : e.k1);
        if (g < 0 || k < 0) continue;
        sv[(size_t)g * p1 + (size_t)k] = e.val;
      }
    }

  std::vector<double> mean, sd;
  if (normalize) {
    standardize(sigma, p1, mean, sd);
    for (size_t i = 0; i < C; i++)
      for (size_t j = 1; j < p1; j++) sv[i * p1 + j] = (sv[i * p1 + j] - mean[j] * sv[i * p1]) / sd[j];
  }
  const size_t p = p1 - 1;
  std::vector<double> S(p * p);
  for (size_t j = 0; j < p; j++)
    for (size_t k = 0; k < p; k++) S[j * p + k] = sigma[(j + 1) * p1 + k + 1];
  std::vector<double> mu(C * p), coef(C * p);  // class means; coef is column-major p x C
  for (size_t i = 0; i < C; i++)
    for (size_t j = 0; j < p; j++) {
      for (size_t k = 0; k < p; k++) S[j * p + k] -= sv[i * p1 + j + 1] * sv[i * p1 + k + 1] / sv[i * p1];
      mu[i * p + j] = coef[i * p + j] = sv[i * p1 + j + 1] / sv[i * p1];
    }
  double tr = 0;
  for (size_t j = 0; j < p; j++) tr += S[j * p + j];
  tr /= (double)p;
  const float keep = 1 - shrinkage;  // float arithmetic, as lda.cpp:268
  for (auto &v : S) v *= keep;
  for (size_t j = 0; j < p; j++) S[j * p + j] += shrinkage * tr;
  for (auto &v : S) v /= t.N;

  solve_symmetric_min_norm(S, p, coef, C);

  std::vector<double> icpt(C);
  for (size_t i = 0; i < C; i++) {
    double d = 0;
    for (size_t j = 0; j < p; j++) d += mu[i * p + j] * coef[i * p + j];
    icpt[i] = -0.5 * d + std::log(sv[i * p1] / t.N);
  }
  if (normalize)
    for (size_t i = 0; i < C; i++)
      for (size_t j = 0; j < p; j++) coef[i * p + j] /= sd[j + 1];

  // [C, #idx, begin offsets of the other key columns + end, their keys, label keys,
  //  coef (class-major), intercepts, (means)]   (lda.cpp:335-386)
  out.clear();
  out.push_back((float)C);
  out.push_back((float)(m == 1 ? 0 : m));
  if (p > n) {
    uint32_t remove = 0;
    for (size_t i = 0; i <= m; i++) {
      if ((int)i == label) { remove = (uint32_t)C; continue; }
      out.push_back((float)(oh.begin[i] - remove));
    }
    for (size_t i = 0; i < lb; i++) out.push_back((float)oh.keys[i]);
    for (size_t i = le; i < oh.keys.size(); i++) out.push_back((float)oh.keys[i]);
  }
  for (size_t i = lb; i < le; i++) out.push_back((float)oh.keys[i]);
  for (double v : coef) out.push_back((float)v);
  for (double v : icpt) out.push_back((float)v);
  if (normalize)
    for (size_t j = 0; j < p; j++) out.push_back((float)mean[j + 1]);
  return true;
}

namespace {
bool as_count(float v, uint64_t hi, uint64_t &out) {
  if (!(v >= 0) || v > (float)hi || v != std::floor(v)) return false;
  out = (uint64_t)v;
  return true;
}
}  // namespace

bool linreg_model(const float *params, uint64_t np, int n_num, int n_cat, bool noise,
                  bool normalize, PredictModel &mdl, std::string &err) {
  uint64_t m = 0, kt = 0;
  if (np < 1 || !as_count(params[0], 1u << 20, m) || (int)m != n_cat) {
    err = "linreg_predict: the parameter vector was trained on another number of key columns";
    return false;
  }
  uint64_t pos = 1;
  mdl = PredictModel{};
  mdl.F = n_num; mdl.M = n_cat; mdl.C = 1;
  mdl.kbegin.assign(1, 0);
  if (m > 0) {
    if (np < 1 + m + 1) { err = "linreg_predict: parameter vector too short"; return false; }
    mdl.kbegin.clear();
    for (uint64_t i = 0; i <= m; i++) {
      uint64_t b = 0;
      if (!as_count(params[pos + i], 1u << 30, b) || (i && b < (uint64_t)mdl.kbegin.back())) {
        err = "linreg_predict: malformed key offsets"; return false;
      }
      mdl.kbegin.push_back((int32_t)b);
    }
    if (mdl.kbegin[0] != 0) { err = "linreg_predict: malformed key offsets"; return false; }
    kt = (uint64_t)mdl.kbegin.back();
    pos += m + 1;
    if (np < pos + kt) { err = "linreg_predict: parameter vector too short"; return false; }
    for (uint64_t i = 0; i < kt; i++) mdl.keys.push_back((int32_t)params[pos + i]);
    pos += kt;
  }
  mdl.KT = (int)kt;
  const uint64_t P = 1 + (uint64_t)n_num + kt;
  const uint64_t want = pos + P + (normalize ? P - 1 : 0) + (noise ? 1 : 0);
  // a vector trained with compute_variance carries one more float than a noise-free predict reads
  if (np != want && !(np == want + 1 && !noise)) {
    err = "linreg_predict: parameter vector length does not match the columns given";
    return false;
  }
  mdl.W.resize(P);
  for (uint64_t i = 0; i < P; i++) mdl.W[i] = (double)params[pos + i];
  if (normalize) {
    const float *mean = params + pos + P;  // numeric means, then one per key
    for (uint64_t i = 1; i < P; i++) mdl.W[0] -= mdl.W[i] * (double)mean[i - 1];
  }
  if (noise) mdl.noise_sd = (double)params[np - 1];
  return true;
}

bool lda_model(const float *params, uint64_t np, int n_num, int n_cat, bool normalize,
               PredictModel &mdl, std::string &err) {
  uint64_t C = 0, nidx = 0, kt = 0;
  if (np < 2 || !as_count(params[0], 1u << 24, C) || C == 0 || !as_count(params[1], 1u << 20, nidx)) {
    err = "lda_predict: malformed parameter vector"; return false;
  }
  if ((nidx == 0 && n_cat != 0) || (nidx > 0 && (int)nidx != n_cat + 1)) {
    err = "lda_predict: the parameter vector was trained on another number of key columns";
    return false;
  }
  mdl = PredictModel{};
  mdl.F = n_num; mdl.M = n_cat; mdl.C = (int)C;
  mdl.kbegin.assign(1, 0);
  uint64_t pos = 2;
  if (nidx > 0) {
    if (np < pos + nidx) { err = "lda_predict: parameter vector too short"; return false; }
    mdl.kbegin.clear();
    for (uint64_t i = 0; i < nidx; i++) {
      uint64_t b = 0;
      if (!as_count(params[pos + i], 1u << 30, b) || (i && b < (uint64_t)mdl.kbegin.back())) {
        err = "lda_predict: malformed key offsets"; return false;
      }
      mdl.kbegin.push_back((int32_t)b);
    }
    kt = (uint64_t)mdl.kbegin.back();
    pos += nidx;
    if (np < pos + kt) { err = "lda_predict: parameter vector too short"; return false; }
    for (uint64_t i = 0; i < kt; i++) mdl.keys.push_back((int32_t)params[pos + i]);
    pos += kt;
  }
  mdl.KT = (int)kt;
  const uint64_t p = (uint64_t)n_num + kt;
  if (np != pos + C + C * p + C + (normalize ? p : 0)) {
    err = "lda_predict: parameter vector length does not match the columns given";
    return false;
  }
  for (uint64_t i = 0; i < C; i++) mdl.labels.push_back((int32_t)params[pos + i]);
  pos += C;
  const float *coef = params + pos, *icpt = coef + C * p, *mean = icpt + C;
  mdl.W.resize(C * (1 + p));
  for (uint64_t k = 0; k < C; k++) {
    double *w = &mdl.W[k * (1 + p)];
    w[0] = (double)icpt[k];
    for (uint64_t j = 0; j < p; j++) {
      w[1 + j] = (double)coef[k * p + j];
      if (normalize) w[0] -= w[1 + j] * (double)mean[j];
    }
  }
  return true;
}

}  // namespace cofactor
