"""GROUP BY pool throughput at one shape (dev tool): python tests/tools/groups_bench.py N G ROWS [is_key]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duckdb-imputation_amd"))
import torch  # noqa: E402

import cofactor_hip  # noqa: E402
from cofactor_hip import ring, synth  # noqa: E402

n, G, rows = int(sys.argv[1]), int(float(sys.argv[2])), int(float(sys.argv[3]))
is_key = len(sys.argv) < 5 or sys.argv[4] != "0"
ctx = cofactor_hip.Context(0)
num, cat = synth.table(torch, 42, n, 0, 0, rows, "cuda", keys=4)
gid = synth.integers(torch, 42, 300, 0, rows, G, "cuda")
grp = ring.Groups(ctx, n, 0, is_key=is_key)
grp.update_device(gid, num, cat)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    grp.update_device(gid, num, cat)
ctx.synchronize()
dt = (time.perf_counter() - t0) / 3
print("groups %d_0 G=%d rows=%d is_key=%d: %.3f ms  %.3e rows/s" % (n, G, rows, is_key, dt * 1e3, rows / dt))
grp.close()
ctx.close()
