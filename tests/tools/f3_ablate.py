"""Dev-only: cost of each phase of fused3_kernel (needs libcofactor_hip_f3dev.so, tests/tools/build_f3dev.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duckdb-imputation_amd"))
os.environ["COFACTOR_LIB"] = os.path.join(ROOT, "duckdb-imputation_amd", "cofactor_hip", "libcofactor_hip_f3dev.so")
os.environ["COFACTOR_FUSED"] = "3"
import torch
import cofactor_hip
rows, n, m, K = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000, 10, 10, 16
masks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 4, 8, 16, 32, 28, 35, 63]
g = torch.Generator(device="cuda").manual_seed(1)
num = [torch.rand(rows, generator=g, device="cuda") for _ in range(n)]
cat = [torch.randint(0, K, (rows,), generator=g, device="cuda", dtype=torch.int32) for _ in range(m)]
torch.cuda.synchronize()
ctx = cofactor_hip.Context(0)
names = {1: "pair products", 2: "make_codes", 4: "pieces", 8: "gram", 16: "S", 32: "whole sum subtile"}
for mask in masks:
    os.environ["COFACTOR_F3_ABLATE"] = str(mask)
    label = "all" if mask == 0 else "without " + " + ".join(v for k, v in names.items() if mask & k)
    agg = ctx.aggregate(n, m)
    agg.update_device(num, cat); ctx.synchronize()
    ctx.profile(True); ctx.profile_read()
    for _ in range(3):
        agg.update_device(num, cat)
    ctx.synchronize()
    p = ctx.profile_read(); ctx.profile(False)
    print("mask %2d  fused3 %.3f ms per %.0e rows  (%s)" % (mask, p["fused_ms"] / max(1, p["fused_launches"]), rows, label), flush=True)
    agg.close()
