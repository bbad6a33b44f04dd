"""Dev probe: repeat small fused2 shapes and count mismatches against the oracle (not a test)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "duckdb-imputation_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ["COFACTOR_FUSED"] = "2"
import torch
import cofactor_hip
from oracle import oracle as orc
from triple_fmt import blob_to_dict
ctx = cofactor_hip.Context(0)
bad = 0
# (the NB shapes run two workgroups per CU; (20, 10), (13, 11), (20, 20) take the sub-launch route)
for (n, m, nb) in [(0, 3, False), (0, 1, False), (3, 2, False), (1, 1, False), (5, 4, False), (10, 10, False), (4, 3, True), (0, 2, True),
                   (2, 5, False), (10, 10, True), (20, 9, True), (20, 10, False), (13, 11, False), (20, 20, False)]:
    rng = np.random.default_rng(1000 + 31 * n + m)
    rows = 200_011 if n + m >= 20 else 20_011
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    cat = [rng.integers(-2, 5, rows).astype(np.int32) for _ in range(m)]
    want = blob_to_dict(orc.State(orc.WIDE).update(num, cat, nb=nb).finalize())
    dn = [torch.from_numpy(c).cuda() for c in num]; dc = [torch.from_numpy(c).cuda() for c in cat]
    torch.cuda.synchronize()
    fails = 0
    for rep in range(40):
        agg = ctx.aggregate(n, m, cofactor_hip.NB if nb else cofactor_hip.TRIPLE)
        agg.update_device(dn, dc)
        got = blob_to_dict(agg.finalize()); agg.close()
        fails += got != want
    print((n, m, nb), "mismatches in 40 runs:", fails, flush=True)
    bad += fails
print("TOTAL", bad)
ctx.close()
