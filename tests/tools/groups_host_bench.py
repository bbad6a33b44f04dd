"""GROUP BY pool fed from host memory in DataChunk-sized batches (the DuckDB glue's pattern), PCIe included
(dev tool): python tests/tools/groups_host_bench.py [N] [G] [ROWS] [CHUNK]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duckdb-imputation_amd"))
import cofactor_hip  # noqa: E402
from cofactor_hip import ring  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
G = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000
rows = int(float(sys.argv[3])) if len(sys.argv) > 3 else 20_000_000
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
rng = np.random.default_rng(1)
gid = rng.integers(0, G, rows).astype(np.int32)
num = [rng.random(rows, dtype=np.float32) for _ in range(n)]
ctx = cofactor_hip.Context(0)
grp = ring.Groups(ctx, n, 0, is_key=False)
L = ring._bind()
ptrs = lambda a: ring._ptr_array([num[k][a:].ctypes.data for k in range(n)])
empty = ring._ptr_array([])
def run():
    for a in range(0, rows, chunk):
        ring._check(L.cofactor_groups_update_host(grp._h, gid[a:].ctypes.data, ptrs(a), empty, min(chunk, rows - a)))
    return grp.count()
run()
t0 = time.perf_counter()
run()
dt = time.perf_counter() - t0
print("groups host path %d_0 G=%d rows=%d chunk=%d: %.1f ms  %.3e rows/s  (%.1f GB/s over the link)" % (n, G, rows, chunk, dt * 1e3, rows / dt, rows * 4 * (n + 1) / dt / 1e9))
grp.close()
ctx.close()
