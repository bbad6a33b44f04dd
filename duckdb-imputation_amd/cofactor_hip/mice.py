"""One MICE run over device-resident columns: the host loop of the
reference's run_MICE_baseline (imputation/algorithms/imputation_base.cpp:6-145) with its three
SQL steps per incomplete column replaced by library calls on columns that stay in HBM:

  SELECT sum_to_triple_n_m(..) FROM t WHERE col_IS_NULL IS FALSE   -> masked aggregate (HIP)
  lda_train / linreg_train(triple, ..)                             -> host fp64, p x p
  CREATE TABLE rep AS SELECT CASE WHEN col_IS_NULL THEN predict(..) ELSE col END + column swap
                                                                   -> predict kernel, in place

Rows are sharded over ranks (one process per GPU); the only exchange is the all-reduce of each
partial triple (dist.allreduce_triple), after which every rank trains the same model.
"""
import time

import numpy as np

from . import dist as cdist
from . import lda_train, linreg_train


class MiceTable:
    """Columns of one shard: float32 numeric and int32 key tensors on the GPU, and for every
    incomplete column a uint8 tensor that is 1 where the value is missing."""

    def __init__(self, num, cat, num_null=None, cat_null=None, first_row=0):
        self.num, self.cat = dict(num), dict(cat)
        self.first_row = int(first_row)      # index of this shard's first row in the whole table
        self.num_null, self.cat_null = dict(num_null or {}), dict(cat_null or {})
        # WHERE col_IS_NULL IS FALSE wants the complement; the masks never change during a run
        self.num_keep = {k: (v == 0).to(v.dtype) for k, v in self.num_null.items()}
        self.cat_keep = {k: (v == 0).to(v.dtype) for k, v in self.cat_null.items()}
        # the library works on its own stream: whatever torch still has queued on these tensors
        # (the two lines above included) must have finished before the first kernel reads them
        import torch
        torch.cuda.synchronize()


def _reduced_triple(agg, dist, device):
    if dist is not None:               # any world size: one rank is how the path is rehearsed
        return cdist.allreduce_triple(agg, dist, device)
    return agg.finalize()


def init_baseline(ctx, table, dist=None, device=None):
    """init_baseline (imputation/algorithms/partition.cpp:671-719): missing numeric values take
    the column's AVG over the present ones, missing keys its MODE (ties: the smallest key)."""
    import torch
    torch.cuda.synchronize(device)                 # torch stream -> library stream hand-over
    for name, keep in table.num_keep.items():
        agg = ctx.aggregate(1, 0)
        agg.update_device_masked([table.num[name]], [], keep)
        b = _reduced_triple(agg, dist, device)
        agg.close()
        mean = float(b[4] / b[3]) if b[3] > 0 else 0.0
        table.num[name].masked_fill_(table.num_null[name].bool(), mean)
        torch.cuda.synchronize(device)
    for name, keep in table.cat_keep.items():
        agg = ctx.aggregate(0, 1)
        agg.update_device_masked([], [table.cat[name]], keep)
        b = _reduced_triple(agg, dist, device)
        agg.close()
        ln = int(b[4])
        keys, counts = b[5:5 + 2 * ln:2], b[6:6 + 2 * ln:2]
        mode = int(keys[int(np.argmax(counts))]) if ln else 0
        table.cat[name].masked_fill_(table.cat_null[name].bool(), mode)
        torch.cuda.synchronize(device)


def run_mice(ctx, table, iterations=1, dist=None, device=None, seed=0, shrinkage=0.001,
             step_size=0.001, max_iterations=10000, timings=None, skip_init=False):
    """run_MICE_baseline: key columns first (LDA, imputation_base.cpp:19-83), then numeric ones
    (stochastic linear regression, :85-142), `iterations` times.  Fills the table in place and
    returns the parameter vectors of the last iteration per column."""
    import torch
    num_names, cat_names = list(table.num), list(table.cat)
    n, m = len(num_names), len(cat_names)
    t_log = timings if timings is not None else {}
    for k in ("aggregate_s", "train_s", "predict_s"):
        t_log.setdefault(k, 0.0)

    def clock():
        torch.cuda.synchronize(device)
        return time.perf_counter()

    if not skip_init:
        init_baseline(ctx, table, dist, device)
    agg = ctx.aggregate(n, m)
    models = {}
    for it in range(iterations):
        for kind, names in (("cat", list(table.cat_null)), ("num", list(table.num_null))):
            for name in names:
                t0 = clock()
                agg.reset()
                keep = (table.cat_keep if kind == "cat" else table.num_keep)[name]
                agg.update_device_masked([table.num[c] for c in num_names],
                                         [table.cat[c] for c in cat_names], keep)
                triple = _reduced_triple(agg, dist, device)
                t1 = clock()
                if kind == "cat":
                    label = cat_names.index(name)
                    params = lda_train(triple, label, shrinkage, False)
                    t2 = clock()
                    ctx.lda_predict(params, [table.num[c] for c in num_names],
                                    [table.cat[c] for c in cat_names if c != name],
                                    out=table.cat[name], mask=table.cat_null[name], emit_label=True)
                else:
                    label = num_names.index(name)
                    params = linreg_train(triple, label, step_size, 0.0, max_iterations, True, False)
                    t2 = clock()
                    # the noise of a row is a function of (seed, GLOBAL row index): the kernel hashes
                    # seed + GOLDEN * (row + 1), so a shard that starts at row `first_row` of the table
                    # passes seed + GOLDEN * first_row and every row draws what it would draw in a
                    # one-GPU run, however the table is sharded (no rank term)
                    col_seed = (seed * 1000003 + it * 10007 + label * 101 + 1) & (2 ** 63 - 1)
                    col_seed = (col_seed + 0x9E3779B97F4A7C15 * table.first_row) & (2 ** 64 - 1)
                    ctx.linreg_predict(params, [table.num[c] for c in num_names if c != name],
                                       [table.cat[c] for c in cat_names], out=table.num[name],
                                       mask=table.num_null[name], noise=True, seed=col_seed)
                t3 = clock()
                models[name] = params
                t_log["aggregate_s"] += t1 - t0
                t_log["train_s"] += t2 - t1
                t_log["predict_s"] += t3 - t2
    agg.close()
    return models
