"""Sparse pair tables (csrc/sparse.hip): column pairs whose dense code-indexed table would be too
big are kept as sorted (key1, key2) -> count lists, like the reference's std::map
(duckdb_extension/src/triple/sum/sum_no_lift.cpp:195-214).  COFACTOR_SPARSE_CELLS lowers the
threshold so that small tables take the path; one test runs it at its real size."""
import numpy as np
import pytest

import cofactor_hip
from oracle import oracle as orc
from triple_fmt import blob_sections, blob_to_dict

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx(monkeypatch):
    monkeypatch.setenv("COFACTOR_SPARSE_CELLS", "4096")      # 128 x 64 codes and beyond: sparse
    c = cofactor_hip.Context(0)
    yield c
    c.close()


def table(rng, rows, n, keys):
    num = [rng.integers(0, 8, rows).astype(np.float32) for _ in range(n)]
    cat = [(rng.integers(0, k, rows) * 13 - 40 * k).astype(np.int32) for k in keys]
    return num, cat


def to_gpu(cols):
    import torch
    out = [torch.from_numpy(np.ascontiguousarray(c)).cuda() for c in cols]
    torch.cuda.synchronize()
    return out


def want_of(num, cat):
    return blob_to_dict(orc.State(orc.FAITHFUL).update(num, cat).finalize())


@pytest.mark.parametrize("keys", [(100, 90), (100, 5, 70), (300, 300, 3, 200)])
def test_sparse_pairs_equal_the_oracle(ctx, keys):
    rng = np.random.default_rng(sum(keys))
    num, cat = table(rng, 30_011, 2, keys)
    agg = ctx.aggregate(2, len(keys))
    agg.update_device(to_gpu(num), to_gpu(cat))
    got = blob_to_dict(agg.finalize())
    agg.close()
    assert got == want_of(num, cat)


def test_batches_growth_from_dense_to_sparse_and_masks(ctx):
    """First batch: few keys (dense tables).  Later batches bring more keys: the pair tables pass
    the threshold and move into their stores with what they hold; one batch is filtered."""
    import torch
    rng = np.random.default_rng(5)
    parts = [table(rng, 5_000, 1, (10, 12)), table(rng, 20_000, 1, (150, 12)), table(rng, 20_000, 1, (150, 140)),
             table(rng, 7_777, 1, (400, 380))]
    keep = (rng.random(20_000) < 0.6).astype(np.uint8)
    agg = ctx.aggregate(1, 2)
    ref = orc.State(orc.FAITHFUL)
    for i, (num, cat) in enumerate(parts):
        if i == 2:
            agg.update_device_masked(to_gpu(num), to_gpu(cat), torch.from_numpy(keep).cuda())
            sel = keep.astype(bool)
            ref.update([c[sel] for c in num], [c[sel] for c in cat])
        else:
            agg.update_device(to_gpu(num), to_gpu(cat))
            ref.update(num, cat)
    got = blob_to_dict(agg.finalize())
    assert got == blob_to_dict(ref.finalize())
    # reset empties the stores too
    agg.reset()
    num, cat = parts[3]
    agg.update_device(to_gpu(num), to_gpu(cat))
    assert blob_to_dict(agg.finalize()) == want_of(num, cat)
    agg.close()


def test_combine_and_host_chunks_with_sparse_pairs(ctx):
    rng = np.random.default_rng(9)
    num, cat = table(rng, 40_000, 2, (200, 180))
    a, b = ctx.aggregate(2, 2), ctx.aggregate(2, 2)
    a.update_host([c[:25_000] for c in num], [c[:25_000] for c in cat])
    b.update_device(to_gpu([c[25_000:] for c in num]), to_gpu([c[25_000:] for c in cat]))
    a.combine(b)
    got = blob_to_dict(a.finalize())
    a.close(); b.close()
    assert got == want_of(num, cat)


def test_table_seam_takes_states_with_sorted_pair_lists(ctx):
    """align_keys / the table image work on a state whose big pair table is a sorted list: the dense
    tables are re-indexed, the list is keyed by the keys themselves and stays as it is; the value of
    the state does not change.  (Across ranks the lists are gathered and merged:
    tests/test_gpu_dist.py::test_two_ranks_with_pair_tables_kept_as_sorted_lists.)"""
    rng = np.random.default_rng(3)
    num, cat = table(rng, 10_000, 1, (200, 180))
    agg = ctx.aggregate(1, 2)
    agg.update_device(to_gpu(num), to_gpu(cat))
    want = want_of(num, cat)
    assert blob_to_dict(agg.finalize()) == want
    keys, offs = agg.keys()
    agg.align_keys(keys, offs)
    assert agg.dict_signature() != 0
    assert any(agg.sparse_is_list(q) for q in range(3)) and int(agg.sparse_lens().sum()) > 0
    assert blob_to_dict(agg.finalize()) == want
    agg.close()


def test_sum_triple_into_a_state_with_sorted_pair_lists(ctx):
    """to_cofactor rows summed (sum_triple) into a state whose wide column pairs are kept as sorted
    lists: their quad_cat entries are packed and merged into the stores (sum.cpp:246-260 adds into a
    std::map); the narrow pair stays a dense table.  Device and host vectors, then plain rows on top."""
    from cofactor_hip import ring
    rng = np.random.default_rng(17)
    num, cat = table(rng, 20_003, 2, (150, 6, 140))
    h = 12_000
    agg = ctx.aggregate(2, 3)
    agg.update_device(to_gpu([c[:4000] for c in num]), to_gpu([c[:4000] for c in cat]))   # the layout: pairs (0,2) ... sparse
    assert any(agg.sparse_is_list(q) for q in range(6)) and not all(agg.sparse_is_list(q) for q in range(6))
    tv = ring.lift_device(ctx, to_gpu([c[4000:h] for c in num]), to_gpu([c[4000:h] for c in cat]))
    ring.update_tvec(agg, tv)
    tvh = ring.lift_host(ctx, [c[h:18_000] for c in num], [c[h:18_000] for c in cat])
    ring.update_tvec(agg, tvh)
    agg.update_device(to_gpu([c[18_000:] for c in num]), to_gpu([c[18_000:] for c in cat]))
    got = blob_to_dict(agg.finalize())
    agg.close()
    assert got == want_of(num, cat)


def test_two_wide_columns_at_real_size():
    """No override: two key columns with 70 000 and 60 000 distinct keys (code capacities 2^17 and
    2^16: 2^33 cells, far past what a dense table may take; also past the 16-bit code cache) next
    to a narrow one, 3e6 rows.  Key counts, per-key sums and all six pair lists against numpy."""
    rng = np.random.default_rng(77)
    rows = 3_000_000
    ctx = cofactor_hip.Context(0)
    num = [rng.integers(0, 8, rows).astype(np.float32)]
    code = [rng.integers(0, 70_000, rows), rng.integers(0, 60_000, rows), rng.integers(0, 5, rows)]
    cat = [(code[0] * 3 - 100_000).astype(np.int32), (code[1] * 5 - 7).astype(np.int32), code[2].astype(np.int32)]
    agg = ctx.aggregate(1, 3)
    half = rows // 2
    agg.update_device(to_gpu([c[:half] for c in num]), to_gpu([c[:half] for c in cat]))
    agg.update_device(to_gpu([c[half:] for c in num]), to_gpu([c[half:] for c in cat]))
    head, lin_cat, num_cat, cat_cat = blob_sections(agg.finalize())
    agg.close(); ctx.close()
    assert head["N"] == rows
    for c in range(3):
        u, cnt = np.unique(cat[c], return_counts=True)
        assert np.array_equal(lin_cat[c][:, 0], u.astype(np.float64))
        assert np.array_equal(lin_cat[c][:, 1], cnt.astype(np.float64))
        s = np.bincount(np.searchsorted(u, cat[c]), weights=num[0].astype(np.float64), minlength=len(u))
        assert np.array_equal(num_cat[c][:, 1], s)
    q = 0
    for c1 in range(3):
        for c2 in range(c1, 3):
            # (signed key1 in the upper half, key2 + 2^31 in the lower: int64 order = (key1, key2) order)
            packed = cat[c1].astype(np.int64) * 2 ** 32 + (cat[c2].astype(np.int64) + 2 ** 31)
            u, cnt = np.unique(packed, return_counts=True)
            got = cat_cat[q]
            assert got.shape[0] == len(u), (c1, c2)
            assert np.array_equal(got[:, 0], (u >> 32).astype(np.float64)), (c1, c2)
            assert np.array_equal(got[:, 1], ((u & 0xFFFFFFFF) - 2 ** 31).astype(np.float64)), (c1, c2)
            assert np.array_equal(got[:, 2], cnt.astype(np.float64)), (c1, c2)
            q += 1
