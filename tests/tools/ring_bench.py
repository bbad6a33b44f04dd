"""Throughput of the batched ring kernels and the GROUP BY pool on one GPU (dev tool; numbers quoted in DESIGN.md).
  python tests/tools/ring_bench.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duckdb-imputation_amd"))
import torch  # noqa: E402

import cofactor_hip  # noqa: E402
from cofactor_hip import ring, synth  # noqa: E402


def timed(fn, ctx, reps=5):
    fn()
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ctx = cofactor_hip.Context(0)
    out = {}
    dev = "cuda"
    # to_cofactor + sum_triple, 4_2 and 10_4
    for n, m, rows in ((4, 2, 20_000_000), (10, 4, 5_000_000)):
        num, cat = synth.table(torch, 42, n, m, 0, rows, dev, keys=8)
        T, nm, Tm = n * (n + 1) // 2, n * m, m * (m + 1) // 2
        wbytes = 4 * (1 + n + T) + 32 + 16 * (3 + m + nm + Tm) + 8 * (m + nm) + 12 * Tm
        tv = ring.lift_device(ctx, num, cat)
        L = ring._bind()
        import ctypes as C
        np_, cp_ = ring._ptr_array([t.data_ptr() for t in num]), ring._ptr_array([t.data_ptr() for t in cat])
        dt = timed(lambda: ring._check(L.cofactor_lift_device(ctx._h, np_, n, cp_, m, rows, 0, C.byref(tv.struct))), ctx)
        out["lift_%d_%d" % (n, m)] = {"rows": rows, "rows_per_s": rows / dt, "written_GBs": rows * wbytes / dt / 1e9, "bytes_per_row": wbytes}
        agg = ctx.aggregate(n, m)
        ring.update_tvec(agg, tv)
        dt = timed(lambda: ring.update_tvec(agg, tv), ctx)
        out["sum_triple_%d_%d" % (n, m)] = {"rows": rows, "rows_per_s": rows / dt, "dense_read_GBs": rows * 4 * (1 + n + T) / dt / 1e9}
        agg.close()
        del tv, num, cat
        torch.cuda.empty_cache()
    # GROUP BY pool: 2_2, 1e5 groups, 2e7 rows; and 20_0 with 1e4 groups
    for n, m, G, rows in ((2, 2, 100_000, 20_000_000), (20, 0, 10_000, 20_000_000), (20, 0, 10_000, 100_000_000),
                          (8, 0, 10_000, 100_000_000)):
        num, cat = synth.table(torch, 42, n, m, 0, rows, dev, keys=4)
        gid = synth.integers(torch, 42, 300, 0, rows, G, dev)
        grp = ring.Groups(ctx, n, m, is_key=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        grp.update_device(gid, num, cat)             # first batch: every group key is new (dictionary growth)
        ctx.synchronize()
        first = time.perf_counter() - t0
        dt = timed(lambda: grp.update_device(gid, num, cat), ctx, reps=3)
        out["groups_%d_%d_G%d_R%.0e" % (n, m, G, rows)] = {"rows": rows, "rows_per_s": rows / dt, "first_batch_s": first}
        if (n, m) == (2, 2):
            A, ka = grp.to_tvec(dev)
            sel = torch.arange(G, device=dev, dtype=torch.int32)
            prod = ring.multiply(ctx, A, A, sel, sel)
            t0 = time.perf_counter()
            for _ in range(3):
                prod = ring.multiply(ctx, A, A, sel, sel)
            torch.cuda.synchronize()
            out["multiply_2_2x2_2"] = {"pairs": G, "pairs_per_s": G / ((time.perf_counter() - t0) / 3)}
            big = sel.repeat(20)                    # 2e6 pairs in one call
            prod = ring.multiply(ctx, A, A, big, big)
            t0 = time.perf_counter()
            for _ in range(3):
                prod = ring.multiply(ctx, A, A, big, big)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            out["multiply_2_2x2_2_big"] = {"pairs": big.numel(), "pairs_per_s": big.numel() / dt}
            # the fill alone into the arrays of the last product (capacities known: no size query, no allocation)
            L = ring._bind()
            import ctypes as C
            call = lambda: ring._check(L.cofactor_multiply_device(ctx._h, C.byref(A.struct), big.data_ptr(), C.byref(A.struct),
                                                                  big.data_ptr(), big.numel(), C.byref(prod.struct), None, None, None))
            dt = timed(call, ctx)
            out["multiply_2_2x2_2_big_one_call"] = {"pairs": big.numel(), "pairs_per_s": big.numel() / dt}
            del prod
        grp.close()
        del num, cat, gid
        torch.cuda.empty_cache()
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
