"""multiply_triple throughput at 2_2 x 2_2 over pool triples (dev tool): python tests/tools/mul_bench.py [PAIRS] [KEYS]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duckdb-imputation_amd"))
import torch  # noqa: E402

import cofactor_hip  # noqa: E402
from cofactor_hip import ring, synth  # noqa: E402

pairs = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
keys = int(sys.argv[2]) if len(sys.argv) > 2 else 4
G, rows = 100_000, 20_000_000
ctx = cofactor_hip.Context(0)
num, cat = synth.table(torch, 42, 2, 2, 0, rows, "cuda", keys=keys)
gid = synth.integers(torch, 42, 300, 0, rows, G, "cuda")
grp = ring.Groups(ctx, 2, 2, is_key=True)
grp.update_device(gid, num, cat)
A, _ = grp.to_tvec("cuda")
sel = torch.arange(G, device="cuda", dtype=torch.int32).repeat((pairs + G - 1) // G)[:pairs].contiguous()
prod = ring.multiply(ctx, A, A, sel, sel)
L = ring._bind()
call = lambda: ring._check(L.cofactor_multiply_device(ctx._h, C.byref(A.struct), sel.data_ptr(), C.byref(A.struct), sel.data_ptr(),
                                                      pairs, C.byref(prod.struct), None, None, None))
call(); ctx.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    call()
ctx.synchronize()
dt = (time.perf_counter() - t0) / 5
s = prod.struct
print("multiply 2_2x2_2 pairs=%d: %.3f ms  %.3e pairs/s  entries lc %d nc %d cc %d" % (pairs, dt * 1e3, pairs / dt, s.lc_cap, s.nc_cap, s.cc_cap))
grp.close()
ctx.close()
