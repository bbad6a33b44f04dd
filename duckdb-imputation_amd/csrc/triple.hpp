// Host-side triple containers of libcofactor_hip: the flat blob codec, the sparse accumulator a
// cofactor_agg keeps for everything that is not (yet) in device tables, and the scalar ring ops.
// No HIP in here.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace cofactor {

inline uint64_t tri(uint64_t k) { return k * (k + 1) / 2; }

// Mirror of one blob: lists exactly as they sit in the blob (see include/cofactor_hip.h).
struct KeyVal { int32_t key; double val; };
struct PairVal { int32_t k1, k2; double val; };
struct ListTriple {
  int kind = 0, n = 0, m = 0;
  double N = 0;
  std::vector<double> lin, quad;
  std::vector<std::vector<KeyVal>> lin_cat;   // m
  std::vector<std::vector<KeyVal>> num_cat;   // n*m (kind 0)
  std::vector<std::vector<PairVal>> cat_cat;  // tri(m) (kind 0)
};

// Walks one blob of at most `cap` doubles; returns its length in doubles (0 if it is malformed or
// does not end within cap: nothing at or beyond b[cap] is read).
uint64_t blob_len(const double *b, uint64_t cap);
bool blob_decode(const double *b, uint64_t cap, ListTriple &t, std::string &err);
void blob_encode(const ListTriple &t, std::vector<double> &out);

// Text form of one triple: DuckDB's STRUCT literal, as Value::ToString() prints the aggregates' result
// and as the MICE drivers paste it back into SQL (imputation/algorithms/imputation_base.cpp:46,116):
//   {'N': 5, 'lin_agg': [15.0, 17.0], 'quad_agg': [...], 'lin_cat': [[{'key': 4, 'value': 3.0}, ..], ..],
//    'quad_num_cat': [[..]], 'quad_cat': [[{'key1': 4, 'key2': 5, 'value': 1.0}, ..], ..]}
// Numbers are printed with 17 significant digits where fewer do not round-trip, so
// text -> triple -> text is the identity.  aggregate_names: lin_agg / quad_agg (aggregates) or
// lin_num / quad_num (scalar functions); the parser takes either, any whitespace, ' or ".
std::string triple_to_text(const ListTriple &t, bool aggregate_names);
bool triple_from_text(const char *text, size_t len, ListTriple &t, std::string &err);

// Sparse accumulator (wide: double sums, exact integer counts up to 2^53).
struct HostTriple {
  int kind = 0, n = 0, m = 0;
  double N = 0;
  std::vector<double> lin, quad;
  // per categorical column: key -> [count, S_0 .. S_{n-1}]   (kind NB: [count])
  std::vector<std::map<int32_t, std::vector<double>>> col;
  // per column pair (c1 <= c2): (key1, key2) -> count        (kind 0 only)
  std::vector<std::map<std::pair<int32_t, int32_t>, double>> pair;

  void shape(int kind_, int n_, int m_);
  void clear();
  // this += one finalised triple (the body of Triple::Sum / sum_nb_agg for one input row)
  bool add_list(const ListTriple &t, std::string &err);
  // this += other (SumStateCombine)
  bool add(const HostTriple &o, std::string &err);
  void encode(std::vector<double> &out) const;  // SumStateFinalize order
  void encode_without_pairs(std::vector<double> &out) const;  // everything before quad_cat
};

// Scalar ring ops on decoded blobs.
void lift_row(const float *const *num, int n, const int32_t *const *cat, int m, uint64_t row,
              int kind, ListTriple &t);
bool multiply(const ListTriple &A, const ListTriple &B, ListTriple &R, std::string &err);
bool add_sub(const ListTriple &A, const ListTriple &B, bool subtract, ListTriple &R,
             std::string &warn);

}  // namespace cofactor
