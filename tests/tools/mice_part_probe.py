"""Where the partitioned MICE variant's aggregate time goes (dev tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duckdb-imputation_amd"))
import torch
import cofactor_hip
from cofactor_hip import mice

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(42)
R, n, m, K = 100_000_000, 10, 10, 16
num = {"x%d" % i: torch.rand(R, device=dev, generator=g) for i in range(n)}
cat = {"k%d" % i: torch.randint(0, K, (R,), device=dev, generator=g, dtype=torch.int32) for i in range(m)}
nulls = lambda: (torch.rand(R, device=dev, generator=g) < 0.1).to(torch.uint8)
t = mice.MiceTable(num, cat, {"x0": nulls(), "x1": nulls()}, {"k0": nulls()})
ctx = cofactor_hip.Context(0)
mice.init_baseline(ctx, t)
pt = mice.PartitionedMiceTable(t)
cols_num = [pt.num[c] for c in num]; cols_cat = [pt.cat[c] for c in cat]
agg = ctx.aggregate(n, m)
agg.update_device(cols_num, cols_cat); agg.finalize()
def sync(): ctx.synchronize(); torch.cuda.synchronize()
for name in pt.names:
    print(name, pt.ranges[name], [(b - a) for a, b in pt.ranges[name]])
    for rep in range(2):
        sync(); t0 = time.perf_counter()
        agg.reset()
        sync(); t1 = time.perf_counter()
        mice._aggregate_ranges(agg, cols_num, cols_cat, pt.ranges[name])
        sync(); t2 = time.perf_counter()
        b = agg.finalize()
        sync(); t3 = time.perf_counter()
        print("  reset %.3f ms  ranges %.3f ms  finalize %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
        for a_, b_ in pt.ranges[name]:
            head = min(b_, (a_ + 3) // 4 * 4)
            for lo, hi in ((a_, head), (head, b_)):
                if hi > lo:
                    sync(); t0 = time.perf_counter()
                    agg.update_device_ptrs([x.data_ptr() + 4 * lo for x in cols_num], [x.data_ptr() + 4 * lo for x in cols_cat], hi - lo)
                    sync(); print("     piece rows %d: %.3f ms" % (hi - lo, (time.perf_counter() - t0) * 1e3))
