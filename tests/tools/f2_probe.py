"""Dev probe: fused2 counts under variations (not a test)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "duckdb-imputation_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import cofactor_hip
from oracle import oracle as orc
from triple_fmt import blob_to_dict

ctx = cofactor_hip.Context(0)

def run(tag, rows, n, m, inf_at=None, nan_at=None, lo=-2, hi=5):
    rng = np.random.default_rng(14)
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    cat = [rng.integers(lo, hi, rows).astype(np.int32) for _ in range(m)]
    if inf_at is not None: num[1][inf_at] = np.inf
    if nan_at is not None: num[2][nan_at] = np.nan
    agg = ctx.aggregate(n, m)
    dn = [torch.from_numpy(c).cuda() for c in num]; dc = [torch.from_numpy(c).cuda() for c in cat]
    torch.cuda.synchronize()
    agg.update_device(dn, dc)
    got = blob_to_dict(agg.finalize()); agg.close()
    want = blob_to_dict(orc.State(orc.WIDE).update(num, cat).finalize())
    ok = [got["lin_cat"][c] == want["lin_cat"][c] for c in range(m)]
    okp = [got["quad_cat"][q] == want["quad_cat"][q] for q in range(len(want["quad_cat"]))]
    print(tag, "lin_cat ok:", ok, "quad_cat ok:", okp, flush=True)
    for c in range(m):
        if not ok[c]:
            print("   col", c, [int(e["value"]) for e in got["lin_cat"][c]], [int(e["value"]) for e in want["lin_cat"][c]])

run("plain 4000", 4000, 3, 2)
run("plain 3840", 3840, 3, 2)
run("inf 4000", 4000, 3, 2, inf_at=17)
run("nan 4000", 4000, 3, 2, nan_at=900)
run("both 4000", 4000, 3, 2, inf_at=17, nan_at=900)
run("both 3840", 3840, 3, 2, inf_at=17, nan_at=900)
run("both 4000 keys>=0", 4000, 3, 2, inf_at=17, nan_at=900, lo=0, hi=7)
run("inf row 300", 4000, 3, 2, inf_at=300)
ctx.close()
