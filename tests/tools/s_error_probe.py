"""Worst realistic case for the fused kernel's fp32 per-key partial sums: ONE key per column, so
every row of a wave lands in the same cell between two fp64 folds.  Prints the largest relative
error of the per-key sums against fp64 sums for a few value distributions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duckdb-imputation_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cofactor_hip
from triple_fmt import blob_to_dict
rows, n, m = 20_000_000, 10, 10
g = torch.Generator(device="cuda").manual_seed(3)
ctx = cofactor_hip.Context(0)
for name, gen in [("uniform[0,1)", lambda: torch.rand(rows, generator=g, device="cuda")),
                  ("lognormal", lambda: torch.exp(3 * torch.randn(rows, generator=g, device="cuda"))),
                  ("1 + tiny", lambda: 1.0 + 1e-3 * torch.rand(rows, generator=g, device="cuda")),
                  ("mixed signs", lambda: torch.randn(rows, generator=g, device="cuda"))]:
    num = [gen().contiguous() for _ in range(n)]
    cat = [torch.zeros(rows, dtype=torch.int32, device="cuda") for _ in range(m)]
    agg = ctx.aggregate(n, m)
    agg.update_device(num, cat)
    t = blob_to_dict(agg.finalize())
    agg.close()
    worst = 0.0
    for k in range(n):
        want = float(num[k].double().sum())
        scale = float(num[k].double().abs().sum())
        got = t["quad_num_cat"][k * m][0]["value"]
        worst = max(worst, abs(got - want) / scale)
    print("%-14s max |S - S_fp64| / sum|x| = %.2e" % (name, worst), flush=True)
