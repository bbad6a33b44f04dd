"""Full-size (BASELINE.json configs) checks through size-independent properties: integer-valued
tables whose exact sums torch can state in int64, and additivity over shards."""
import numpy as np
import pytest

import cofactor_hip
from triple_fmt import blob_to_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = cofactor_hip.Context(0)
    yield c
    c.close()


def test_20_0_100M_rows_exact_on_integer_table(ctx):
    """C2 size (1e8 rows x 20 float columns = 8 GB).  Values in 0..7, so every sum is an integer
    below 2^53: the HIP result must equal torch's int64 sums exactly."""
    import torch
    rows, n = 100_000_000, 20
    g = torch.Generator(device="cuda").manual_seed(42)
    cols = [torch.randint(0, 8, (rows,), generator=g, device="cuda", dtype=torch.int32).float()
            for _ in range(n)]
    torch.cuda.synchronize()
    agg = ctx.aggregate(n, 0)
    agg.update_device(cols, [])
    got = blob_to_dict(agg.finalize())
    assert got["N"] == rows
    ints = [c.to(torch.int32) for c in cols]
    lin = [int(c.sum(dtype=torch.int64)) for c in ints]
    assert got["lin_agg"] == [float(v) for v in lin]
    q = 0
    for j in range(n):
        for k in range(j, n):
            if (j * 7 + k) % 5 == 0 or j == k:           # a spread of 70-odd pairs
                want = int((ints[j] * ints[k]).sum(dtype=torch.int64))
                assert got["quad_agg"][q] == float(want), (j, k)
            q += 1
    # additivity: two shards sum to the whole, bit for bit on this table
    a, b = ctx.aggregate(n, 0), ctx.aggregate(n, 0)
    cut = 37_000_001
    a.update_device_ptrs([c.data_ptr() for c in cols], [], cut)
    b.update_device_ptrs([c.data_ptr() + 4 * cut for c in cols], [], rows - cut)   # 4-B aligned only
    a.combine(b)
    assert blob_to_dict(a.finalize()) == got
    for x in (agg, a, b):
        x.close()


def test_10_10_100M_rows_counts_exact(ctx):
    """C3 size (1e8 rows, 10 float + 10 int32 columns, 16 keys per column): key counts and pair
    counts must equal torch.bincount exactly; per-key sums of 0..7-valued columns are exact too."""
    import torch
    rows, n, m, K = 100_000_000, 10, 10, 16
    g = torch.Generator(device="cuda").manual_seed(7)
    num = [torch.randint(0, 8, (rows,), generator=g, device="cuda", dtype=torch.int32).float()
           for _ in range(n)]
    cat = [torch.randint(0, K, (rows,), generator=g, device="cuda", dtype=torch.int32) for _ in range(m)]
    torch.cuda.synchronize()
    agg = ctx.aggregate(n, m)
    agg.update_device(num, cat)
    got = blob_to_dict(agg.finalize())
    agg.close()
    assert got["N"] == rows
    for c in range(m):
        cnt = torch.bincount(cat[c], minlength=K).cpu().tolist()
        assert [e["key"] for e in got["lin_cat"][c]] == list(range(K))
        assert [e["value"] for e in got["lin_cat"][c]] == [float(v) for v in cnt]
    for (k, c) in [(0, 0), (3, 7), (9, 9), (5, 2)]:
        s = torch.bincount(cat[c], weights=num[k].double(), minlength=K).cpu().tolist()
        assert [e["value"] for e in got["quad_num_cat"][k * m + c]] == s
    q = 0
    for c1 in range(m):
        for c2 in range(c1, m):
            if (c1 + 3 * c2) % 4 == 0:
                pc = torch.bincount(cat[c1].long() * K + cat[c2].long(), minlength=K * K).cpu().tolist()
                want = [(a, b, float(pc[a * K + b])) for a in range(K) for b in range(K) if pc[a * K + b]]
                assert [(e["key1"], e["key2"], e["value"]) for e in got["quad_cat"][q]] == want
            q += 1


def test_10_10_thousand_keys_per_column_exact(ctx):
    """SURVEY §8(d)'s stress input: 10 float + 10 int32 columns with K = 1000 keys each, 4e7 rows
    (pair tables of 1e6 cells each, all 5.5e7 cells non-empty: the generic path's HBM pair
    launches and finalize's threaded list encoding).  Key counts, per-key sums of 0..7-valued
    columns and every pair table against torch.bincount, exactly; keys are spread out and
    negative so that codes differ from keys."""
    import torch
    from triple_fmt import blob_sections
    rows, n, m, K = 40_000_000, 10, 10, 1000
    g = torch.Generator(device="cuda").manual_seed(11)
    num = [torch.randint(0, 8, (rows,), generator=g, device="cuda", dtype=torch.int32).float() for _ in range(n)]
    code = [torch.randint(0, K, (rows,), generator=g, device="cuda", dtype=torch.int32) for _ in range(m)]
    cat = [(c * 37 - 5000).contiguous() for c in code]
    torch.cuda.synchronize()
    agg = ctx.aggregate(n, m)
    agg.update_device(num, cat)
    blob = agg.finalize()
    agg.close()
    head, lin_cat, num_cat, cat_cat = blob_sections(blob)
    assert head["N"] == rows
    keys = np.arange(K, dtype=np.float64) * 37 - 5000
    for c in range(m):
        cnt = torch.bincount(code[c], minlength=K).cpu().numpy().astype(np.float64)
        assert np.array_equal(lin_cat[c][:, 0], keys)
        assert np.array_equal(lin_cat[c][:, 1], cnt)
    for (k, c) in [(0, 0), (3, 7), (9, 9), (5, 2)]:
        s = torch.bincount(code[c], weights=num[k].double(), minlength=K).cpu().numpy()
        assert np.array_equal(num_cat[k * m + c][:, 0], keys)
        assert np.array_equal(num_cat[k * m + c][:, 1], s)
    q = 0
    for c1 in range(m):
        for c2 in range(c1, m):
            pc = torch.bincount(code[c1].long() * K + code[c2].long(), minlength=K * K).cpu().numpy()
            nz = np.flatnonzero(pc)
            got = cat_cat[q]
            assert got.shape[0] == nz.size, (c1, c2)
            assert np.array_equal(got[:, 0], keys[nz // K]), (c1, c2)
            assert np.array_equal(got[:, 1], keys[nz % K]), (c1, c2)
            assert np.array_equal(got[:, 2], pc[nz].astype(np.float64)), (c1, c2)
            q += 1


@pytest.mark.parametrize("env", ["COFACTOR_GRAM_DMA", "COFACTOR_GRAM_RING"])
@pytest.mark.parametrize("n", [1, 4, 7, 12, 13, 20])
def test_lds_dma_variant_of_the_dense_kernel_is_exact(monkeypatch, n, env):
    """COFACTOR_GRAM_DMA=1 sends whole tiles through gram_dma_kernel (tiles fetched by LDS-DMA, 4 or 8
    waves per tile), COFACTOR_GRAM_RING=1 through gram_ring_kernel (dedicated loader waves), the tail
    through gram_kernel: same integer-valued table, same exact sums."""
    import torch
    monkeypatch.setenv(env, "1")
    c = cofactor_hip.Context(0)
    rows = 3_000_000 + 77
    g = torch.Generator(device="cuda").manual_seed(300 + n)
    ints = [torch.randint(0, 8, (rows,), generator=g, device="cuda", dtype=torch.int32) for _ in range(n)]
    cols = [x.float() for x in ints]
    torch.cuda.synchronize()
    agg = c.aggregate(n, 0)
    agg.update_device(cols, [])
    got = blob_to_dict(agg.finalize())
    agg.close(); c.close()
    assert got["N"] == rows
    assert got["lin_agg"] == [float(int(x.sum(dtype=torch.int64))) for x in ints]
    q = 0
    for j in range(n):
        for k in range(j, n):
            assert got["quad_agg"][q] == float(int((ints[j] * ints[k]).sum(dtype=torch.int64))), (j, k)
            q += 1


@pytest.mark.parametrize("n", [1, 3, 4, 7, 8, 10, 12, 13, 16])
def test_narrow_tables_exact_through_the_tile_ring(ctx, n):
    """Narrow tables run gram_kernel with a ring of several tiles per workgroup and several rows
    per MFMA; the ring only reaches its steady state when a workgroup walks many tiles, i.e. above
    about 1.6 M rows.  Integer-valued columns, ragged row count, with and without a row filter:
    every sum must equal torch's int64 sums exactly."""
    import torch
    rows = 6_000_000 + 777
    g = torch.Generator(device="cuda").manual_seed(100 + n)
    ints = [torch.randint(0, 8, (rows,), generator=g, device="cuda", dtype=torch.int32) for _ in range(n)]
    cols = [c.float() for c in ints]
    mask = (torch.rand(rows, generator=g, device="cuda") < 0.7).to(torch.uint8)
    torch.cuda.synchronize()
    for keep in (None, mask):
        agg = ctx.aggregate(n, 0)
        if keep is None:
            agg.update_device(cols, [])
            sel = ints
        else:
            agg.update_device_masked(cols, [], keep)
            sel = [c * keep.to(torch.int32) for c in ints]
        got = blob_to_dict(agg.finalize())
        agg.close()
        assert got["N"] == (rows if keep is None else int(keep.sum()))
        assert got["lin_agg"] == [float(int(c.sum(dtype=torch.int64))) for c in sel]
        q = 0
        for j in range(n):
            for k in range(j, n):
                want = int((sel[j] * ints[k]).sum(dtype=torch.int64))
                assert got["quad_agg"][q] == float(want), (j, k)
                q += 1


@pytest.mark.parametrize("n", [4, 10, 14, 16, 20])
def test_unaligned_columns_exact_through_the_tile_ring(ctx, n):
    """Same as above for columns that are only 4-byte aligned (dword loads instead of dwordx4)."""
    import torch
    rows = 6_000_000 + 333
    g = torch.Generator(device="cuda").manual_seed(200 + n)
    ints = [torch.randint(0, 8, (rows + 1,), generator=g, device="cuda", dtype=torch.int32) for _ in range(n)]
    cols = [c.float() for c in ints]
    torch.cuda.synchronize()
    agg = ctx.aggregate(n, 0)
    agg.update_device_ptrs([c.data_ptr() + 4 for c in cols], [], rows)      # skip each column's first row
    got = blob_to_dict(agg.finalize())
    agg.close()
    sel = [c[1:] for c in ints]
    assert got["N"] == rows
    assert got["lin_agg"] == [float(int(c.sum(dtype=torch.int64))) for c in sel]
    q = 0
    for j in range(n):
        for k in range(j, n):
            if j == k or (j + k) % 3 == 0:
                assert got["quad_agg"][q] == float(int((sel[j] * sel[k]).sum(dtype=torch.int64))), (j, k)
            q += 1


def test_20_0_one_billion_rows_exact_on_integer_table(ctx):
    """C4's table on one GPU (BASELINE.json configs[3]: 1e9 rows x 20 float columns = 80 GB).
    Whole numbers 0..15 from the shard-reproducible generator (cofactor_hip/synth.py), so every sum
    is an integer below 2^53 and the HIP result must equal torch's int64 sums exactly: N, all 20
    lin_agg entries and 70-odd quad_agg entries."""
    import torch
    from cofactor_hip import synth
    rows, n = 1_000_000_000, 20
    cols, _ = synth.table(torch, 42, n, 0, 0, rows, "cuda", exact=16)
    torch.cuda.synchronize()
    agg = ctx.aggregate(n, 0)
    agg.update_device(cols, [])
    got = blob_to_dict(agg.finalize())
    agg.close()
    assert got["N"] == rows
    step = 1 << 27
    lin = [0] * n
    picked = [(j, k) for j in range(n) for k in range(j, n) if (j * 7 + k) % 3 == 0 or j == k]
    quad = {p: 0 for p in picked}
    for a in range(0, rows, step):
        ints = [c[a:a + step].to(torch.int64) for c in cols]
        for k in range(n):
            lin[k] += int(ints[k].sum())
        for (j, k) in picked:
            quad[(j, k)] += int((ints[j] * ints[k]).sum())
        del ints
    assert got["lin_agg"] == [float(v) for v in lin]
    assert len(picked) >= 70
    q = 0
    for j in range(n):
        for k in range(j, n):
            if (j, k) in quad:
                assert got["quad_agg"][q] == float(quad[(j, k)]), (j, k)
            q += 1
    # the first rows of the table are what the host-side generator says they are
    h, _ = synth.table_np(42, n, 0, 0, 1000, exact=16)
    assert all(np.array_equal(c[:1000].cpu().numpy(), hc) for c, hc in zip(cols, h))


def test_group_by_20_0_segmented_at_150M_rows_exact(ctx):
    """GROUP BY of sum_to_triple_20_0 through the segmented path (groupseg.hip) at a size that needs two
    sub-batches (> 2^27 rows), key-typed groups, twice (the second call takes the counting pass as the
    dictionary check): every group's N, and lin / a spread of quad cells of some groups, against torch
    int64 sums over the group's rows."""
    import torch
    from cofactor_hip import ring
    rows, n, G = 150_000_000, 20, 1000
    g = torch.Generator(device="cuda").manual_seed(7)
    cols = [torch.randint(0, 8, (rows,), generator=g, device="cuda", dtype=torch.int32).float() for _ in range(n)]
    slot = torch.randint(0, G, (rows,), generator=g, device="cuda", dtype=torch.int32)
    gid = (slot * 37 - 5000).to(torch.int32)
    torch.cuda.synchronize()
    grp = ring.Groups(ctx, n, 0, is_key=True)
    grp.update_device(gid, cols, [])
    grp.update_device(gid, cols, [])
    assert grp.count() == G
    counts = torch.bincount(slot.long(), minlength=G)
    for s_ in (0, 1, 499, 999):
        got = blob_to_dict(grp.finalize(s_ * 37 - 5000))
        sel = slot == s_
        assert got["N"] == 2 * int(counts[s_])
        ints = [c[sel].to(torch.int64) for c in cols]
        assert got["lin_agg"] == [float(2 * int(v.sum())) for v in ints]
        q = 0
        for j in range(n):
            for k in range(j, n):
                if (j * 7 + k) % 11 == 0 or j == k:
                    assert got["quad_agg"][q] == float(2 * int((ints[j] * ints[k]).sum())), (s_, j, k)
                q += 1
    grp.close()
