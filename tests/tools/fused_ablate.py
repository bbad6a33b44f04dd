"""Dev-only: cost of each phase of fused_kernel (needs libcofactor_hip_dev.so)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duckdb-imputation_amd"))
os.environ["COFACTOR_LIB"] = os.path.join(ROOT, "duckdb-imputation_amd", "cofactor_hip", "libcofactor_hip_dev.so")
import torch
import cofactor_hip
rows, n, m, K = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000, 10, 10, 16
g = torch.Generator(device="cuda").manual_seed(1)
num = [torch.rand(rows, generator=g, device="cuda") for _ in range(n)]
cat = [torch.randint(0, K, (rows,), generator=g, device="cuda", dtype=torch.int32) for _ in range(m)]
torch.cuda.synchronize()
ctx = cofactor_hip.Context(0)
for mask, label in [(0, "all"), (1, "no phase2 atomics"), (2, "no S mfma"), (4, "no gram"), (8, "no lookups"),
                    (16, "no pieces"), (31, "loads+park only"), (30, "phase2 only"), (29, "S mfma only")]:
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
    print("%-20s mask %2d: fused %.2f ms (launches %d)  whole update %.2f ms -> %.2e rows/s"
          % (label, mask, p["fused_ms"] / 3, p["fused_launches"], dt * 1e3, rows / dt), flush=True)
    agg.close()
