"""Throughput of the DuckDB-facing host path (cofactor_agg_update_host): 2048-row chunks of host
columns -> pinned staging -> H2D -> HIP kernels.  PCIe-inclusive; reported in DESIGN.md, never as
bench.py's `value`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duckdb-imputation_amd"))
import numpy as np
import cofactor_hip

rows, n, m = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000, 20, 0
if len(sys.argv) > 3:
    n, m = int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(0)
num = [rng.random(rows, dtype=np.float32) for _ in range(n)]
cat = [rng.integers(0, 16, rows).astype(np.int32) for _ in range(m)]
ctx = cofactor_hip.Context(0)
for chunk in (2048, 1 << 20):
    agg = ctx.aggregate(n, m)
    t0 = time.perf_counter()
    for lo in range(0, rows, chunk):
        hi = min(rows, lo + chunk)
        agg.update_host([c[lo:hi] for c in num], [c[lo:hi] for c in cat])
    blob = agg.finalize()
    dt = time.perf_counter() - t0
    assert int(blob[3]) == rows
    print("update_host %d_%d, %d rows in %d-row chunks: %.3f s  %.3e rows/s  %.2f GB/s of input"
          % (n, m, rows, chunk, dt, rows / dt, rows * 4 * (n + m) / dt / 1e9), flush=True)
    agg.close()
