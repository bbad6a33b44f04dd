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


# ---- the partitioned variant ---------------------------------------------------------------------------
# imputation/algorithms/imputation_low.cpp / imputation_high.cpp keep the table PARTITIONED by its null
# pattern, so that "the rows where col is missing" are whole partitions, and keep one cofactor of the
# whole table up to date instead of aggregating it again for every column:
#     triple(rows where col is present) = triple(all rows) - triple(rows where col is missing)
# Per column that is two aggregates over the ~10 % of rows where it is missing (before and after the
# imputation), contiguous row ranges of the reordered table — no row filter, nothing read of the rows
# without a missing value.

def _gray_rank(p):
    """Place of bit pattern p in the reflected Gray sequence: patterns are laid out in that order, so
    the partitions where a given column is missing form one or two runs instead of up to 2^(k-1)."""
    b = p
    shift = 1
    while (p >> shift) > 0:
        b ^= p >> shift
        shift += 1
    return b


class PartitionedMiceTable:
    """A MiceTable's rows reordered by null pattern (stable within a pattern).  `ranges[name]` are the
    row ranges [a, b) of the reordered columns in which `name` is missing; `row_ids` the rows' places
    in the original shard (the imputation noise is a function of those)."""

    def __init__(self, table):
        import torch
        self.source = table
        self.names = list(table.cat_null) + list(table.num_null)          # processing order = bit order
        k = len(self.names)
        assert 1 <= k <= 8
        nulls = [table.cat_null[c] if c in table.cat_null else table.num_null[c] for c in self.names]
        rows = nulls[0].numel()
        pat = torch.zeros(rows, dtype=torch.int32, device=nulls[0].device)
        for b, nl in enumerate(nulls):
            pat += (nl != 0).to(torch.int32) << b
        rank_of = torch.tensor([_gray_rank(p) for p in range(1 << k)], dtype=torch.int32, device=pat.device)
        rank = rank_of[pat.long()]
        order = torch.argsort(rank, stable=True)
        self.order = order
        self.row_ids = order.to(torch.int32).contiguous()                  # (< 2^31 rows per shard; read as uint32)
        self.num = {c: v[order].contiguous() for c, v in table.num.items()}
        self.cat = {c: v[order].contiguous() for c, v in table.cat.items()}
        counts = torch.bincount(rank.long(), minlength=1 << k).cpu().numpy()
        start = np.concatenate([[0], np.cumsum(counts)])
        pattern_at = [0] * (1 << k)
        for p in range(1 << k):
            pattern_at[_gray_rank(p)] = p
        self.ranges = {}
        for b, name in enumerate(self.names):
            runs = []
            for r in range(1 << k):
                if (pattern_at[r] >> b) & 1 and counts[r] > 0:
                    a, e = int(start[r]), int(start[r + 1])
                    if runs and runs[-1][1] == a:
                        runs[-1] = (runs[-1][0], e)
                    else:
                        runs.append((a, e))
            self.ranges[name] = runs
        torch.cuda.synchronize()

    def write_back(self):
        """The imputed columns back into the source table's row order."""
        import torch
        for name in self.names:
            src = self.cat[name] if name in self.cat and name in self.source.cat_null else self.num[name]
            dst = self.source.cat[name] if name in self.source.cat_null else self.source.num[name]
            dst[self.order] = src
        torch.cuda.synchronize()


def _aggregate_ranges(agg, cols_num, cols_cat, ranges):
    """agg += the rows of the ranges.  A range that does not start on a multiple of four rows is fed in
    two pieces (the library's fast kernels want 16-byte aligned columns)."""
    for a, b in ranges:
        head = min(b, (a + 3) // 4 * 4)
        for lo, hi in ((a, head), (head, b)):
            if hi > lo:
                agg.update_device_ptrs([t.data_ptr() + 4 * lo for t in cols_num], [t.data_ptr() + 4 * lo for t in cols_cat], hi - lo)


def run_mice_partitioned(ctx, table, iterations=1, dist=None, device=None, seed=0, shrinkage=0.001,
                         step_size=0.001, max_iterations=10000, timings=None, skip_init=False, part=None):
    """run_mice over the table partitioned by null pattern.  Same models and — to the rounding of one
    blob subtraction — the same imputed values as run_mice; returns (models, partitioned table).  The
    source table is written when `part.write_back()` is called."""
    import torch
    from . import add as triple_add
    from . import sub as triple_sub
    num_names, cat_names = list(table.num), list(table.cat)
    n, m = len(num_names), len(cat_names)
    t_log = timings if timings is not None else {}
    for k in ("aggregate_s", "train_s", "predict_s", "partition_s"):
        t_log.setdefault(k, 0.0)

    def clock():
        torch.cuda.synchronize(device)
        return time.perf_counter()

    if not skip_init and part is None:
        init_baseline(ctx, table, dist, device)
    t0 = clock()
    pt = part if part is not None else PartitionedMiceTable(table)
    cols_num = [pt.num[c] for c in num_names]
    cols_cat = [pt.cat[c] for c in cat_names]
    agg = ctx.aggregate(n, m)
    if getattr(pt, "total", None) is None:
        agg.update_device(cols_num, cols_cat)                    # the cofactor of the whole (filled) table, once
        pt.total = _reduced_triple(agg, dist, device)
    t_log["partition_s"] += clock() - t0
    models = {}
    for it in range(iterations):
        for name in pt.names:
            kind = "cat" if name in table.cat_null else "num"
            ranges = pt.ranges[name]
            t0 = clock()
            agg.reset()
            _aggregate_ranges(agg, cols_num, cols_cat, ranges)
            old = _reduced_triple(agg, dist, device)
            triple = triple_sub(pt.total, old)                   # the rows where `name` is present
            t1 = clock()
            if kind == "cat":
                label = cat_names.index(name)
                params = lda_train(triple, label, shrinkage, False)
                t2 = clock()
                for a, b in ranges:
                    ctx.lda_predict(params, [t[a:b] for t in cols_num],
                                    [pt.cat[c][a:b] for c in cat_names if c != name],
                                    out=pt.cat[name][a:b], mask=None, emit_label=True)
            else:
                label = num_names.index(name)
                params = linreg_train(triple, label, step_size, 0.0, max_iterations, True, False)
                t2 = clock()
                col_seed = (seed * 1000003 + it * 10007 + label * 101 + 1) & (2 ** 63 - 1)
                col_seed = (col_seed + 0x9E3779B97F4A7C15 * table.first_row) & (2 ** 64 - 1)
                for a, b in ranges:
                    ctx.linreg_predict(params, [pt.num[c][a:b] for c in num_names if c != name],
                                       [t[a:b] for t in cols_cat], out=pt.num[name][a:b], mask=None,
                                       noise=True, seed=col_seed, row_ids=pt.row_ids[a:b])
            t3 = clock()
            agg.reset()
            _aggregate_ranges(agg, cols_num, cols_cat, ranges)
            new = _reduced_triple(agg, dist, device)
            pt.total = triple_add(triple, new)
            t4 = clock()
            models[name] = params
            t_log["aggregate_s"] += (t1 - t0) + (t4 - t3)
            t_log["train_s"] += t2 - t1
            t_log["predict_s"] += t3 - t2
    agg.close()
    return models, pt
