// AddressSanitizer / UBSan pass over the host-only translation units of libcofactor_hip
// (triple.cpp: blob codec and ring ops; ml.cpp: trainers and parameter-vector parsing), CPU build.
// GPU sanitizers are not available on the pool, so the device code is covered by parity tests
// only.  Input: a file of concatenated flat triple blobs written by tests/test_host_sanitize.py:
//   u64 count, then per blob: u64 length, doubles.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "ml.hpp"
#include "triple.hpp"

using namespace cofactor;

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "host_sanitize: %s failed at line %d\n", #c, __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
  REQUIRE(argc == 2);
  FILE *f = fopen(argv[1], "rb");
  REQUIRE(f);
  uint64_t count = 0;
  REQUIRE(fread(&count, 8, 1, f) == 1);
  std::vector<std::vector<double>> blobs(count);
  for (auto &b : blobs) {
    uint64_t len = 0;
    REQUIRE(fread(&len, 8, 1, f) == 1);
    b.resize(len);
    REQUIRE(fread(b.data(), 8, len, f) == len);
  }
  fclose(f);
  std::string err;
  size_t trained = 0, multiplied = 0;
  for (auto const &b : blobs) {
    REQUIRE(blob_len(b.data(), b.size()) == b.size());
    ListTriple t;
    REQUIRE(blob_decode(b.data(), b.size(), t, err));
    // every proper prefix is rejected and never read past (heap copies of exactly that size:
    // ASan sees any read beyond them); a lying list header as well
    for (size_t cut : {size_t(0), size_t(3), b.size() / 3, b.size() / 2, b.size() - 1}) {
      std::vector<double> pre(b.begin(), b.begin() + cut);
      REQUIRE(blob_len(pre.data(), pre.size()) == 0);
      ListTriple junk;
      REQUIRE(!blob_decode(pre.data(), pre.size(), junk, err));
    }
    if (t.m > 0) {
      std::vector<double> lying(b);
      lying[4 + t.n + (t.kind ? t.n : t.n * (t.n + 1) / 2)] = 1e9;
      REQUIRE(blob_len(lying.data(), lying.size()) == 0);
    }
    std::vector<double> again;
    blob_encode(t, again);
    REQUIRE(again == b);
    HostTriple h;
    h.shape(t.kind, t.n, t.m);
    REQUIRE(h.add_list(t, err));
    REQUIRE(h.add_list(t, err));
    HostTriple h2;
    h2.shape(t.kind, t.n, t.m);
    REQUIRE(h2.add(h, err));
    std::vector<double> enc;
    h2.encode(enc);
    REQUIRE(enc[3] == 2 * t.N);
    ListTriple sum, diff;
    std::string warn;
    add_sub(t, t, false, sum, warn);
    add_sub(sum, t, true, diff, warn);
    REQUIRE(diff.N == t.N);
    for (auto const &o : blobs) {                   // every same-kind pair goes through multiply
      ListTriple u, r;
      REQUIRE(blob_decode(o.data(), o.size(), u, err));
      if (u.kind != t.kind) continue;
      REQUIRE(multiply(t, u, r, err));
      REQUIRE(r.n == t.n + u.n && r.m == t.m + u.m);
      multiplied++;
    }
    if (t.kind == 0 && t.N > 10) {
      std::vector<float> params;
      PredictModel mdl;
      for (int label = 0; label < t.n; label++)
        for (int flags = 0; flags < 4; flags++) {
          REQUIRE(linreg_train(t, label, 0.001f, flags & 1 ? 0.1f : 0.f, 300, flags & 2, flags & 1, params, err));
          REQUIRE(linreg_model(params.data(), params.size(), t.n - 1, t.m, flags & 2, flags & 1, mdl, err));
          // truncated / padded vectors must be rejected, not read past
          std::vector<float> cut(params.begin(), params.begin() + params.size() / 2);
          REQUIRE(!linreg_model(cut.data(), cut.size(), t.n - 1, t.m, false, false, mdl, err));
          trained++;
        }
      for (int label = 0; label < t.m; label++)
        for (int norm = 0; norm < 2; norm++) {
          if (t.n + t.m < 2) continue;
          REQUIRE(lda_train(t, label, norm ? 0.f : 0.05f, norm, params, err));
          REQUIRE(lda_model(params.data(), params.size(), t.n, t.m - 1, norm, mdl, err));
          std::vector<float> cut(params.begin(), params.begin() + params.size() / 2);
          REQUIRE(!lda_model(cut.data(), cut.size(), t.n, t.m - 1, norm, mdl, err));
          trained++;
        }
    }
  }
  // malformed headers never get walked
  const double bad[8] = {0, -1, 2, 0, 0, 0, 0, 0};
  REQUIRE(blob_len(bad, 8) == 0);
  const double bad2[8] = {0, 0, 1, 5, -3, 0, 0, 0};
  REQUIRE(blob_len(bad2, 8) == 0);
  printf("host_sanitize ok: %zu blobs, %zu products, %zu models\n", blobs.size(), multiplied, trained);
  return 0;
}
