// Consumers of a finalised triple: ridge linear regression and shrinkage LDA
// trained from the cofactor matrix alone, in fp64 on the host, emitting the reference's flat
// FLOAT[] parameter vectors.  No HIP in here; the per-row predict kernels are in predict.hip.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "triple.hpp"

namespace cofactor {

// One-hot column layout of a triple: for key column c the sorted distinct keys are
// keys[begin[c] .. begin[c+1])  (n_cols_1hot_expansion with drop_first = 0, ML/utils.cpp:520-575).
struct OneHot {
  std::vector<uint32_t> begin;  // m + 1
  std::vector<int64_t> keys;
  size_t width(int n) const { return 1 + (size_t)n + keys.size(); }
};
void onehot_layout(const ListTriple &t, OneHot &oh);

// sigma = [1, x, onehot(keys)]^T [1, x, onehot(keys)] (row-major p x p), key column `skip_cat`
// left out when >= 0 (build_sigma_matrix, ML/utils.cpp:176-310).  Returns p.
size_t build_sigma(const ListTriple &t, const OneHot &oh, int skip_cat, std::vector<double> &sigma);

// Minimum-norm least squares solve of the symmetric system A X = B (A p x p row-major, destroyed;
// B p x nrhs column-major, overwritten by X) — what dgelsd(rcond = -1) gives the reference
// (ML/lda.cpp:294-297).  Cholesky when A is safely positive definite, cyclic Jacobi otherwise.
void solve_symmetric_min_norm(std::vector<double> &A, size_t p, std::vector<double> &B, size_t nrhs);

// linreg_train(triple, label, step_size, lambda, max_iterations, compute_variance, normalize)
// (ML::ridge_linear_regression, ML/regression.cpp:108-354).
bool linreg_train(const ListTriple &t, int label, float step_size, float lambda, int max_iterations,
                  bool compute_variance, bool normalize, std::vector<float> &params,
                  std::string &err);

// lda_train(triple, label, shrinkage, normalize)  (lda_train, ML/lda.cpp:161-416).
bool lda_train(const ListTriple &t, int label, float shrinkage, bool normalize,
               std::vector<float> &params, std::string &err);

// What the predict kernel needs, unpacked from a parameter vector: out_k = W[k] . [1, x, onehot]
// with W row-major C x (1 + F + KT); a standardised model's means are folded into W[k][0].
struct PredictModel {
  int F = 0, M = 0, C = 0, KT = 0;
  std::vector<int32_t> kbegin, keys, labels;
  std::vector<double> W;
  double noise_sd = 0;
};
// layouts read back as ML::linreg_impute (regression.cpp:424-436) / LDA_impute (lda.cpp:448-500) do
bool linreg_model(const float *params, uint64_t n_params, int n_num, int n_cat, bool noise,
                  bool normalize, PredictModel &mdl, std::string &err);
bool lda_model(const float *params, uint64_t n_params, int n_num, int n_cat, bool normalize,
               PredictModel &mdl, std::string &err);

}  // namespace cofactor
