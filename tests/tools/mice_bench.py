"""Config C5 at one GPU: seconds per MICE iteration on a synthetic n_m table with
10 % of the entries of two numeric and one key column missing, columns resident in HBM.
  python tests/tools/mice_bench.py [--rows 100000000] [--num-cols 10] [--cat-cols 10] [--keys 16]
Prints one JSON line with the per-phase split (aggregate / train / predict)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duckdb-imputation_amd"))

import torch  # noqa: E402

import cofactor_hip  # noqa: E402
from cofactor_hip import mice  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--num-cols", type=int, default=10)
    ap.add_argument("--cat-cols", type=int, default=10)
    ap.add_argument("--keys", type=int, default=16)
    ap.add_argument("--iterations", type=int, default=2)
    ap.add_argument("--partitioned", action="store_true", help="run_mice_partitioned: rows reordered by null pattern")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(42)
    R, n, m, K = a.rows, a.num_cols, a.cat_cols, a.keys
    num = {"x%d" % i: torch.rand(R, device=dev, generator=g) for i in range(n)}
    cat = {"k%d" % i: torch.randint(0, K, (R,), device=dev, generator=g, dtype=torch.int32) for i in range(m)}
    # make the incomplete columns depend on the others, so the models have something to learn
    num["x0"] = (0.6 * num["x2"] - 0.3 * num["x3"] + 0.05 * cat["k1"].float() + 0.1 * torch.randn(R, device=dev, generator=g)).contiguous()
    num["x1"] = (num["x4"] * 0.5 + 0.02 * cat["k2"].float() + 0.1 * torch.randn(R, device=dev, generator=g)).contiguous()
    cat["k0"] = ((num["x5"] * K * 0.5 + cat["k3"].float() * 0.5 + torch.rand(R, device=dev, generator=g)).to(torch.int32) % K).contiguous()
    nulls = lambda: (torch.rand(R, device=dev, generator=g) < 0.1).to(torch.uint8)
    t = mice.MiceTable(num, cat, {"x0": nulls(), "x1": nulls()}, {"k0": nulls()})
    ctx = cofactor_hip.Context(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mice.init_baseline(ctx, t)
    torch.cuda.synchronize()
    init_s = time.perf_counter() - t0
    log = {}
    if a.partitioned:
        setup = {}
        _, part = mice.run_mice_partitioned(ctx, t, iterations=1, skip_init=True, timings=setup)   # warm-up + the reordering
        t0 = time.perf_counter()
        mice.run_mice_partitioned(ctx, t, iterations=a.iterations, seed=1, timings=log, skip_init=True, part=part)
        torch.cuda.synchronize()
        per_it = (time.perf_counter() - t0) / a.iterations
        log["partition_once_s"] = setup["partition_s"]
    else:
        mice.run_mice(ctx, t, iterations=1, skip_init=True)        # warm-up iteration (dictionaries, LDS plans)
        t0 = time.perf_counter()
        mice.run_mice(ctx, t, iterations=a.iterations, seed=1, timings=log, skip_init=True)
        torch.cuda.synchronize()
        per_it = (time.perf_counter() - t0) / a.iterations
    print(json.dumps({"metric": "seconds per MICE iteration (2 numeric + 1 key column imputed)",
                      "value": per_it, "unit": "s", "rows": R, "shape": "%d_%d" % (n, m), "keys": K,
                      "init_baseline_s": init_s,
                      "aggregate_s": log["aggregate_s"] / a.iterations,
                      "train_s": log["train_s"] / a.iterations,
                      "predict_s": log["predict_s"] / a.iterations,
                      "variant": "partitioned by null pattern" if a.partitioned else "row filter",
                      "partition_once_s": log.get("partition_once_s")}))


if __name__ == "__main__":
    main()
