"""Dev-only timing experiment: cost of each part of cat_accumulate (needs libcofactor_hip_dev.so,
built with -DCOFACTOR_DEV_ABLATE; results are wrong whenever a mask bit is set)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duckdb-imputation_amd"))
os.environ["COFACTOR_LIB"] = os.path.join(ROOT, "duckdb-imputation_amd", "cofactor_hip", "libcofactor_hip_dev.so")
import torch
import cofactor_hip

rows, n, m = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000, 10, 10
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
os.environ.setdefault("COFACTOR_NO_FUSED", "1")      # this tool is about the two-kernel path
g = torch.Generator(device="cuda").manual_seed(1)
num = [torch.rand(rows, generator=g, device="cuda") for _ in range(n)]
cat = [torch.randint(0, K, (rows,), generator=g, device="cuda", dtype=torch.int32) for _ in range(m)]
torch.cuda.synchronize()
ctx = cofactor_hip.Context(0)
for mask, label in [(0, "all"), (1, "no cnt"), (2, "no S"), (4, "no pairs"), (6, "cnt only"), (7, "lookups only"),
                    (8, "S as f32"), (5, "S only"), (13, "S only f32"), (3, "pairs only")]:
    os.environ["COFACTOR_CAT_ABLATE"] = str(mask)
    agg = ctx.aggregate(n, m)
    agg.update_device(num, cat); ctx.synchronize()
    ctx.profile(True); ctx.profile_read()
    t0 = time.perf_counter()
    for _ in range(3):
        agg.update_device(num, cat)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 3
    p = ctx.profile_read(); ctx.profile(False)
    print("%-14s mask %2d: cat_accumulate %.2f ms  gram %.2f ms  whole update %.2f ms  -> %.2e rows/s"
          % (label, mask, p["cat_ms"] / 3, p["gram_ms"] / 3, dt * 1e3, rows / dt), flush=True)
    agg.close()
