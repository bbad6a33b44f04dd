// Throughput of the DuckDB-facing host path measured from C++ (no interpreter between the calls):
// cofactor_agg_update_host over 2048-row chunks of host columns, as DuckDB's executor delivers them.
//   g++ -O2 -std=c++17 -Iinclude tests/tools/host_path_bench.cpp -Lduckdb-imputation_amd/cofactor_hip \
//       -lcofactor_hip -Wl,-rpath,'$ORIGIN/../../duckdb-imputation_amd/cofactor_hip' -o tests/tools/host_path_bench
//   tests/tools/host_path_bench [rows] [n] [m] [threads]
// PCIe-inclusive; reported in DESIGN.md, never as bench.py's `value`.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "cofactor_hip.h"

int main(int argc, char **argv) {
  const uint64_t rows = argc > 1 ? (uint64_t)atof(argv[1]) : 20000000;
  const int n = argc > 2 ? atoi(argv[2]) : 20, m = argc > 3 ? atoi(argv[3]) : 0;
  const int threads = argc > 4 ? atoi(argv[4]) : 1;
  std::vector<std::vector<float>> num(n, std::vector<float>(rows));
  std::vector<std::vector<int32_t>> cat(m, std::vector<int32_t>(rows));
  std::mt19937 rng(1);
  for (auto &c : num) for (auto &v : c) v = (float)(rng() >> 8) * (1.0f / 16777216.0f);
  for (auto &c : cat) for (auto &v : c) v = (int32_t)(rng() & 15);
  cofactor_ctx *ctx = nullptr;
  if (cofactor_ctx_create(0, &ctx) != COFACTOR_OK) { fprintf(stderr, "%s\n", cofactor_last_error()); return 1; }
  // every configuration twice: the second pass is what the next query sees (staging blocks come
  // from the context's pool instead of being pinned again)
  for (uint64_t chunk : {(uint64_t)2048, (uint64_t)2048, (uint64_t)1 << 20, (uint64_t)1 << 20}) {
    std::vector<cofactor_agg *> aggs(threads, nullptr);
    for (auto &a : aggs) {                     // warm-up: staging buffers, dictionaries, code objects
      cofactor_agg_create(ctx, n, m, COFACTOR_TRIPLE, &a);
      std::vector<const float *> np(n);
      std::vector<const int32_t *> cp(m);
      for (int k = 0; k < n; k++) np[k] = num[k].data();
      for (int c = 0; c < m; c++) cp[c] = cat[c].data();
      uint64_t need = 0;
      cofactor_agg_update_host(a, np.data(), cp.data(), nullptr, nullptr, nullptr, std::min<uint64_t>(rows, 4096));
      cofactor_agg_finalize(a, nullptr, 0, &need);
      cofactor_agg_reset(a);
    }
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
      pool.emplace_back([&, t] {            // one thread-local state per worker, like DuckDB
        const uint64_t lo = rows * t / threads, hi = rows * (t + 1) / threads;
        std::vector<const float *> np(n);
        std::vector<const int32_t *> cp(m);
        for (uint64_t r = lo; r < hi; r += chunk) {
          const uint64_t take = std::min(chunk, hi - r);
          for (int k = 0; k < n; k++) np[k] = num[k].data() + r;
          for (int c = 0; c < m; c++) cp[c] = cat[c].data() + r;
          if (cofactor_agg_update_host(aggs[t], np.data(), cp.data(), nullptr, nullptr, nullptr, take) != COFACTOR_OK) {
            fprintf(stderr, "%s\n", cofactor_last_error());
            exit(1);
          }
        }
      });
    for (auto &th : pool) th.join();
    for (int t = 1; t < threads; t++) cofactor_agg_combine(aggs[0], aggs[t]);
    uint64_t need = 0;
    cofactor_agg_finalize(aggs[0], nullptr, 0, &need);
    std::vector<double> blob(need);
    cofactor_agg_finalize(aggs[0], blob.data(), need, &need);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("update_host %d_%d, %llu rows, %llu-row chunks, %d thread(s): %.3f s  %.3e rows/s  %.2f GB/s of input (N=%.0f)\n",
           n, m, (unsigned long long)rows, (unsigned long long)chunk, threads, dt, rows / dt,
           rows * 4.0 * (n + m) / dt / 1e9, blob[3]);
    for (auto &a : aggs) cofactor_agg_destroy(a);
  }
  cofactor_ctx_destroy(ctx);
  return 0;
}
