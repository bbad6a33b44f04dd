"""Dev-only: where does gram_kernel's time go (needs libcofactor_hip_dev.so)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duckdb-imputation_amd"))
os.environ["COFACTOR_LIB"] = os.path.join(ROOT, "duckdb-imputation_amd", "cofactor_hip", "libcofactor_hip_dev.so")
import torch
import cofactor_hip
rows, n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000, 20
g = torch.Generator(device="cuda").manual_seed(1)
num = [torch.rand(rows, generator=g, device="cuda") for _ in range(n)]
torch.cuda.synchronize()
for wgs in (4,):
    os.environ["COFACTOR_GRAM_WGS_PER_CU"] = str(wgs)
    ctx = cofactor_hip.Context(0)
    for mask, label in [(0, "full"), (1, "loads + park + barriers, no MFMA"), (2, "loads + barriers"), (3, "bare loads")]:
        os.environ["COFACTOR_GRAM_ABLATE"] = str(mask)
        agg = ctx.aggregate(n, 0)
        agg.update_device(num, []); ctx.synchronize()
        ctx.profile(True); ctx.profile_read()
        for _ in range(5):
            agg.update_device(num, [])
        p = ctx.profile_read(); ctx.profile(False)
        ms = p["gram_ms"] / 5
        print("wgs/CU %d %-28s %.3f ms  %.0f GB/s" % (wgs, label, ms, 4 * n * rows / ms / 1e6), flush=True)
        agg.close()
    ctx.close()
