"""cofactor_agg_combine on the device (Triple::SumStateCombine, sum_state.cpp:10-114): states of two
CONTEXTS (what a DuckDB process with one context per GPU merges; on a one-GPU box both contexts sit
on device 0, the code path — event hand-over, export, import(add), list merge — is the same), key
sets that overlap, are disjoint or nested, NB kind, filtered updates, host-side addends, pair tables
kept as sorted lists.  Every result is compared with the oracle over all rows."""
import os
import time

import numpy as np
import pytest

import cofactor_hip
from oracle import oracle as orc
from triple_fmt import blob_to_dict

pytestmark = pytest.mark.gpu


_KEEP = []        # update_device is asynchronous: the columns must outlive the kernels that read them


def _dev(cols):
    import torch
    out = [torch.from_numpy(np.ascontiguousarray(c)).cuda() for c in cols]
    torch.cuda.synchronize()
    _KEEP.append(out)
    return out


@pytest.fixture(autouse=True)
def _drop_columns():
    yield
    import torch
    torch.cuda.synchronize()
    _KEEP.clear()


def _table(rng, rows, n, m, lo, hi):
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    cat = [rng.integers(lo, hi + 2 * c, rows).astype(np.int32) for c in range(m)]
    return num, cat


@pytest.mark.parametrize("n,m,nb,ranges", [
    (20, 0, False, ((0, 1), (0, 1))),
    (3, 2, False, ((-3, 9), (5, 20))),            # overlapping key sets
    (10, 10, False, ((0, 16), (0, 16))),          # the one-pass kernels on both sides
    (4, 3, False, ((0, 8), (100, 140))),          # disjoint key sets
    (2, 2, False, ((0, 40), (10, 20))),           # src's keys nested in dst's
    (5, 2, True, ((0, 9), (4, 30))),              # NB kind
    (0, 3, False, ((-5, 5), (0, 70))),
])
def test_combine_of_two_contexts_equals_the_whole_table(n, m, nb, ranges):
    rng = np.random.default_rng(7 * n + m)
    rows_a, rows_b = 70_001, 50_003
    num_a, cat_a = _table(rng, rows_a, n, m, *ranges[0])
    num_b, cat_b = _table(rng, rows_b, n, m, *ranges[1])
    kind = cofactor_hip.NB if nb else cofactor_hip.TRIPLE
    c1, c2 = cofactor_hip.Context(0), cofactor_hip.Context(0)
    a, b = c1.aggregate(n, m, kind), c2.aggregate(n, m, kind)
    a.update_device(_dev(num_a), _dev(cat_a))
    b.update_device(_dev(num_b), _dev(cat_b))
    want_b = blob_to_dict(orc.State(orc.WIDE).update(num_b, cat_b, nb=nb).finalize())
    a.combine(b)
    whole = orc.State(orc.WIDE).update(num_a, cat_a, nb=nb).update(num_b, cat_b, nb=nb)
    assert blob_to_dict(a.finalize()) == blob_to_dict(whole.finalize())
    assert blob_to_dict(b.finalize()) == want_b                     # src keeps its value
    # again: both states now share one alignment (no key exchange), and dst keeps accumulating
    a.combine(b)
    a.update_device(_dev(num_b), _dev(cat_b))
    whole.update(num_b, cat_b, nb=nb).update(num_b, cat_b, nb=nb)
    assert blob_to_dict(a.finalize()) == blob_to_dict(whole.finalize())
    for x in (a, b):
        x.close()
    c1.close(); c2.close()


def test_combine_into_an_empty_state_and_of_an_empty_state():
    rng = np.random.default_rng(3)
    num, cat = _table(rng, 30_000, 3, 2, 0, 12)
    ctx = cofactor_hip.Context(0)
    full, empty, other = ctx.aggregate(3, 2), ctx.aggregate(3, 2), ctx.aggregate(3, 2)
    full.update_device(_dev(num), _dev(cat))
    want = blob_to_dict(orc.State(orc.WIDE).update(num, cat).finalize())
    empty.combine(full)                                             # lazily shaped from src (sum_state.cpp:26-60)
    assert blob_to_dict(empty.finalize()) == want
    full.combine(other)                                             # nothing to add
    assert blob_to_dict(full.finalize()) == want
    for x in (full, empty, other):
        x.close()
    ctx.close()


def test_combine_with_filtered_updates_and_host_side_triples():
    """dst and src hold rows kept by a filter (device counter), and src also holds lifted triples on
    the host (update_triples): everything must arrive."""
    import torch
    rng = np.random.default_rng(11)
    rows, n, m = 40_000, 4, 2
    num, cat = _table(rng, rows, n, m, -2, 11)
    mask = (rng.random(rows) < 0.5).astype(np.uint8)
    extra_num, extra_cat = _table(rng, 9, n, m, 50, 55)             # keys no device table has seen
    c1, c2 = cofactor_hip.Context(0), cofactor_hip.Context(0)
    a, b = c1.aggregate(n, m), c2.aggregate(n, m)
    dm = torch.from_numpy(mask).cuda()
    a.update_device_masked(_dev(num), _dev(cat), dm)
    b.update_device_masked(_dev(num), _dev(cat), dm)
    b.update_triples(cofactor_hip.lift_host(extra_num, extra_cat))
    a.combine(b)
    sel = mask.astype(bool)
    whole = orc.State(orc.WIDE)
    for _ in range(2):
        whole.update([c[sel] for c in num], [c[sel] for c in cat])
    whole.update(extra_num, extra_cat)
    assert blob_to_dict(a.finalize()) == blob_to_dict(whole.finalize())
    a.close(); b.close(); c1.close(); c2.close()


def test_combine_merges_sorted_pair_lists(monkeypatch):
    """With the threshold lowered the pair tables of these columns are sorted lists: combine merges
    them (sort + reduce by key), also when dst's table is still dense and src's is a list."""
    monkeypatch.setenv("COFACTOR_SPARSE_CELLS", "2000")
    rng = np.random.default_rng(5)
    rows, n, m = 60_000, 2, 2
    num_a = [rng.integers(0, 9, rows).astype(np.float32) for _ in range(n)]
    cat_a = [rng.integers(0, 20, rows).astype(np.int32) for _ in range(m)]          # 20 x 20: dense
    num_b = [rng.integers(0, 9, rows).astype(np.float32) for _ in range(n)]
    cat_b = [rng.integers(-50, 150, rows).astype(np.int32) for _ in range(m)]       # 200 x 200 > 2000 cells: lists
    c1, c2 = cofactor_hip.Context(0), cofactor_hip.Context(0)
    a, b = c1.aggregate(n, m), c2.aggregate(n, m)
    a.update_device(_dev(num_a), _dev(cat_a))
    b.update_device(_dev(num_b), _dev(cat_b))
    assert int(b.sparse_lens().sum()) > 0
    a.combine(b)
    assert int(a.sparse_lens().sum()) > 0
    whole = orc.State(orc.WIDE).update(num_a, cat_a).update(num_b, cat_b)
    assert blob_to_dict(a.finalize()) == blob_to_dict(whole.finalize())
    b.combine(a)                                                    # list into list
    whole.update(num_b, cat_b)
    assert blob_to_dict(b.finalize()) == blob_to_dict(whole.finalize())
    a.close(); b.close(); c1.close(); c2.close()


def test_combine_at_1000_keys_per_column_is_exact_and_stays_on_the_device():
    """K = 1000 (SURVEY.md §8d's stress shape): 10 key columns, 55 pair tables of 1 M cells each.
    The merge is remap + export + import(add) kernels; counts are checked against bincounts and the
    time is printed (combine time at K = 1000)."""
    import torch
    rows, n, m, K = 2_000_000, 2, 10, 1000
    g = torch.Generator(device="cuda").manual_seed(9)
    c1, c2 = cofactor_hip.Context(0), cofactor_hip.Context(0)
    parts = []
    for ctx in (c1, c2):
        num = [torch.randint(0, 8, (rows,), generator=g, device="cuda").float() for _ in range(n)]
        cat = [torch.randint(0, K, (rows,), generator=g, device="cuda", dtype=torch.int32) for _ in range(m)]
        torch.cuda.synchronize()
        agg = ctx.aggregate(n, m)
        agg.update_device(num, cat)
        parts.append((agg, num, cat))
    (a, num_a, cat_a), (b, num_b, cat_b) = parts
    c1.synchronize(); c2.synchronize()
    t0 = time.perf_counter()
    a.combine(b)
    c1.synchronize()
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    a.combine(b)                                                    # same alignment: no key exchange, no remap
    c1.synchronize()
    again = time.perf_counter() - t0
    print("combine at K=1000, 10 key columns: %.1f ms (first, with alignment), %.1f ms (aligned)" % (first * 1e3, again * 1e3))
    got = blob_to_dict(a.finalize())
    assert got["N"] == 3 * rows
    cat_all = [torch.cat([x, y, y]) for x, y in zip(cat_a, cat_b)]
    num_all = [torch.cat([x, y, y]) for x, y in zip(num_a, num_b)]
    for c in (0, 9):
        cnt = torch.bincount(cat_all[c].long(), minlength=K).cpu().numpy()
        assert [kv["value"] for kv in got["lin_cat"][c]] == [float(v) for v in cnt if v]
        s0 = torch.bincount(cat_all[c].long(), weights=num_all[0].double(), minlength=K).cpu().numpy()
        assert np.allclose([kv["value"] for kv in got["quad_num_cat"][0 * m + c]], s0[cnt > 0], rtol=0, atol=0)
    pc = torch.bincount(cat_all[0].long() * K + cat_all[9].long(), minlength=K * K).cpu().numpy()
    q = 9                                                           # pair (0, 9) in upper-triangle order
    assert [e["value"] for e in got["quad_cat"][q]] == [float(v) for v in pc if v]
    for x in (a, b):
        x.close()
    c1.close(); c2.close()


def test_library_communicator_with_one_rank():
    """cofactor_comm_* / cofactor_agg_allreduce (csrc/comm.cpp) with a world of one: RCCL is opened by
    the library, the id / init / all-gather / all-reduce / import chain runs on the context stream
    and leaves the state's value unchanged (the N > 1 run is tests/test_gpu_dist.py's two-GPU test
    and the driver's scaling bench)."""
    rng = np.random.default_rng(2)
    ctx = cofactor_hip.Context(0)
    comm = cofactor_hip.Comm(ctx, cofactor_hip.comm_unique_id(), 0, 1)
    for (n, m, nb) in [(20, 0, False), (3, 2, False), (10, 10, False), (4, 2, True)]:
        num, cat = _table(rng, 50_000, n, m, -3, 13)
        agg = ctx.aggregate(n, m, cofactor_hip.NB if nb else cofactor_hip.TRIPLE)
        agg.update_device(_dev(num), _dev(cat))
        want = blob_to_dict(orc.State(orc.WIDE).update(num, cat, nb=nb).finalize())
        agg.allreduce(comm)
        assert blob_to_dict(agg.finalize()) == want
        agg.allreduce(comm)                                         # aligned now: no key exchange
        assert blob_to_dict(agg.finalize()) == want
        agg.close()
    comm.close()
    ctx.close()
