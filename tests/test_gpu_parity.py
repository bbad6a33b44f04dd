"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the reference's
golden vectors.  Bars: N, keys and key counts bit-exact; lin/quad and per-key sums bit-exact on
integer-valued inputs, within 1e-5 relative of the fp64 oracle on random floats
(BASELINE.json north_star)."""
import numpy as np
import pytest

import cofactor_hip
from golden_cases import cases
from oracle import oracle as orc
from triple_fmt import assert_triple_close, blob_to_dict, dense_truth

pytestmark = pytest.mark.gpu

RTOL = 1e-5        # north_star: "triple values within 1e-5 relative of CPU"


@pytest.fixture(scope="module")
def ctx():
    c = cofactor_hip.Context(0)
    yield c
    c.close()


def gpu_triple(ctx, num, cat, nb=False, via="device"):
    import torch
    agg = ctx.aggregate(len(num), len(cat), cofactor_hip.NB if nb else cofactor_hip.TRIPLE)
    if via == "device":
        dn = [torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)).cuda() for c in num]
        dc = [torch.from_numpy(np.ascontiguousarray(c, dtype=np.int32)).cuda() for c in cat]
        torch.cuda.synchronize()
        agg.update_device(dn, dc)
    else:
        agg.update_host(num, cat)
    blob = agg.finalize()
    agg.close()
    return blob


def int_table(rng, rows, n, m, k=7):
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    cat = [rng.integers(-2, k - 2, rows).astype(np.int32) for _ in range(m)]
    return num, cat


# ---- the reference's own known-answer tests, through the HIP path --------------------------
class GpuBackend:
    def __init__(self, ctx):
        self.ctx = ctx

    def sum_to(self, num, cat, nb):
        return gpu_triple(self.ctx, num, cat, nb, via="host")      # DataChunk path -> HIP kernels

    def lift(self, num, cat, nb):
        return cofactor_hip.lift_host(num, cat, cofactor_hip.NB if nb else cofactor_hip.TRIPLE)

    def sum_lifted(self, blobs, nb):
        n, m = int(blobs[0][1]), int(blobs[0][2])
        agg = self.ctx.aggregate(n, m, cofactor_hip.NB if nb else cofactor_hip.TRIPLE)
        agg.update_triples(blobs)
        out = agg.finalize()
        agg.close()
        return out

    def multiply(self, a, b, nb):
        return cofactor_hip.multiply(a, b)


def test_reference_goldens_through_hip(ctx, goldens, ref_table):
    for name, fn in cases(goldens, ref_table):
        for got, want in fn(GpuBackend(ctx)):
            assert got == want, name


# ---- dense Gram kernel: exact on integer data, every n, ragged row counts -------------------
@pytest.mark.parametrize("n", list(range(1, 21)))
def test_dense_exact_every_n(ctx, n):
    rng = np.random.default_rng(100 + n)
    rows = 3001 + 17 * n                        # not a multiple of the 256-row tile
    num, _ = int_table(rng, rows, n, 0)
    # asymmetric data: column j is scaled differently so a transposed block cannot pass
    num = [np.minimum(c + j, 31).astype(np.float32) for j, c in enumerate(num)]
    got = blob_to_dict(gpu_triple(ctx, num, []))
    want = blob_to_dict(orc.State(orc.FAITHFUL).update(num, []).finalize())
    assert got == want


@pytest.mark.parametrize("rows", [1, 2, 3, 4, 5, 63, 64, 255, 256, 257, 511, 513, 1024, 70001])
def test_dense_exact_ragged_rows(ctx, rows):
    rng = np.random.default_rng(rows)
    num, _ = int_table(rng, rows, 20, 0)
    got = blob_to_dict(gpu_triple(ctx, num, []))
    want = blob_to_dict(orc.State(orc.FAITHFUL).update(num, []).finalize())
    assert got == want


def test_dense_unaligned_columns(ctx):
    """Column pointers that are only 4-byte aligned take the scalar-load variant."""
    import torch
    rng = np.random.default_rng(5)
    rows, n = 5003, 7
    base = [rng.integers(0, 16, rows + 3).astype(np.float32) for _ in range(n)]
    dev = [torch.from_numpy(b).cuda() for b in base]
    views = [d[1 + (j % 3):1 + (j % 3) + rows] for j, d in enumerate(dev)]
    torch.cuda.synchronize()
    agg = ctx.aggregate(n, 0)
    agg.update_device_ptrs([v.data_ptr() for v in views], [], rows)
    got = blob_to_dict(agg.finalize())
    agg.close()
    host = [b[1 + (j % 3):1 + (j % 3) + rows] for j, b in enumerate(base)]
    assert got == blob_to_dict(orc.State(orc.FAITHFUL).update(host, []).finalize())


def test_dense_random_floats_within_tolerance(ctx):
    rng = np.random.default_rng(42)
    rows, n = 1_000_000, 20
    num = [rng.random(rows, dtype=np.float32) for _ in range(n)]
    got = blob_to_dict(gpu_triple(ctx, num, []))
    want = blob_to_dict(orc.State(orc.WIDE).update(num, [], threads=8).finalize())
    assert_triple_close(got, want, rtol=RTOL)
    # and the oracle itself agrees with an independent numpy/fp64 statement
    assert_triple_close(want, dense_truth(num, []), rtol=1e-9)


def test_config_c1_4_0_one_million_rows(ctx):
    """BASELINE.json configs[0]: sum_to_triple_4_0 over a 1M-row x 4 float-column table (the
    reference's own CPU-runnable case), HIP path vs both oracle modes."""
    rng = np.random.default_rng(4)
    rows, n = 1_000_000, 4
    num = [rng.random(rows, dtype=np.float32) for _ in range(n)]
    got = blob_to_dict(gpu_triple(ctx, num, []))
    wide = blob_to_dict(orc.State(orc.WIDE).update(num, []).finalize())
    assert_triple_close(got, wide, rtol=RTOL)
    # the reference's fp32 running sums are themselves only ~1e-4 accurate here; the HIP result
    # must sit well inside their error band around the exact value
    faithful = blob_to_dict(orc.State(orc.FAITHFUL).update(num, []).finalize())
    for key in ("lin_agg", "quad_agg"):
        g, w, f = (np.array(d[key]) for d in (got, wide, faithful))
        assert np.all(np.abs(g - w) <= np.abs(f - w) + 1e-6 * np.abs(w))


def test_non_finite_inputs_propagate_like_the_reference(ctx):
    """inf / nan in a numeric column: sums touching them become inf / nan exactly where the
    oracle's do (dense path and the fused kernel's bf16-piece path); everything else is exact."""
    rng = np.random.default_rng(14)
    rows = 4000
    num, cat = int_table(rng, rows, 3, 2)
    num[1][17] = np.inf
    num[2][900] = np.nan
    for c in ([], cat):
        got = blob_to_dict(gpu_triple(ctx, num, c))
        want = blob_to_dict(orc.State(orc.WIDE).update(num, c).finalize())
        for key in ["lin_agg", "quad_agg"]:
            np.testing.assert_array_equal(np.array(got[key]), np.array(want[key]))
        if c:
            assert got["lin_cat"] == want["lin_cat"] and got["quad_cat"] == want["quad_cat"]
            for g, w in zip(got["quad_num_cat"], want["quad_num_cat"]):
                np.testing.assert_array_equal(np.array([e["value"] for e in g]), np.array([e["value"] for e in w]))


def test_dense_large_magnitudes_and_signs(ctx):
    rng = np.random.default_rng(9)
    rows, n = 200_000, 5
    num = [((rng.random(rows) - 0.5) * 10.0 ** (j - 1)).astype(np.float32) for j in range(n)]
    got = blob_to_dict(gpu_triple(ctx, num, []))
    want = blob_to_dict(orc.State(orc.WIDE).update(num, []).finalize())
    # cancelling sums: tolerance relative to the sum of magnitudes
    mag = dense_truth([np.abs(c) for c in num], [])
    for key in ("lin_agg", "quad_agg"):
        g, w, s = (np.array(d[key]) for d in (got, want, mag))
        assert np.all(np.abs(g - w) <= RTOL * s + 1e-30), key


# ---- categorical kernels -----------------------------------------------------------------------
@pytest.mark.parametrize("n,m", [(0, 1), (0, 3), (1, 1), (3, 3), (10, 10), (20, 20), (2, 5)])
def test_mixed_exact_on_integer_data(ctx, n, m):
    rng = np.random.default_rng(1000 + 31 * n + m)
    rows = 20_011
    num, cat = int_table(rng, rows, n, m)
    got = blob_to_dict(gpu_triple(ctx, num, cat))
    want = blob_to_dict(orc.State(orc.FAITHFUL).update(num, cat).finalize())
    assert got == want


def test_extreme_and_negative_keys(ctx):
    rng = np.random.default_rng(3)
    rows = 10_000
    pool = np.array([-2**31, 2**31 - 1, 0, -1, 1, 123456789, -987654321], dtype=np.int32)
    cat = [pool[rng.integers(0, len(pool), rows)] for _ in range(3)]
    num = [rng.integers(0, 8, rows).astype(np.float32) for _ in range(2)]
    got = blob_to_dict(gpu_triple(ctx, num, cat))
    want = blob_to_dict(orc.State(orc.FAITHFUL).update(num, cat).finalize())
    assert got == want


def test_many_keys_grow_dictionary_and_tables(ctx):
    """600 distinct keys: dictionaries (64 slots) and code tables (16 codes) must grow, and the
    tables no longer fit LDS, so the global-atomic variant runs."""
    rng = np.random.default_rng(11)
    rows = 50_000
    cat = [(rng.integers(0, 600, rows) * 7919 - 1000).astype(np.int32), rng.integers(0, 40, rows).astype(np.int32)]
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(3)]
    got = blob_to_dict(gpu_triple(ctx, num, cat))
    want = blob_to_dict(orc.State(orc.FAITHFUL).update(num, cat).finalize())
    assert got == want


def test_keys_arriving_in_later_batches(ctx):
    """Streaming updates where each batch brings unseen keys: accumulated tables are re-laid out
    without losing counts."""
    import torch
    rng = np.random.default_rng(12)
    n, m = 2, 2
    agg = ctx.aggregate(n, m)
    ref = orc.State(orc.FAITHFUL)
    for batch in range(6):
        rows = 4000 + batch
        hi = 5 * (batch + 1) ** 2                       # 5, 20, 45, ... distinct keys
        num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
        cat = [rng.integers(0, hi, rows).astype(np.int32) for _ in range(m)]
        dn = [torch.from_numpy(c).cuda() for c in num]
        dc = [torch.from_numpy(c).cuda() for c in cat]
        torch.cuda.synchronize()
        agg.update_device(dn, dc)
        ref.update(num, cat)
        assert blob_to_dict(agg.finalize()) == blob_to_dict(ref.finalize())   # finalize is repeatable
    agg.close()


def test_optimistic_pass_redoes_tiles_with_new_keys(ctx):
    """Second and later updates skip the dictionary pass; tiles that meet a key the dictionary
    does not know are left out by the fused kernel and redone.  Batches: known keys only, a few
    new keys in a few tiles, new keys everywhere (still <= 16 keys), then > 16 keys."""
    import torch
    rng = np.random.default_rng(99)
    n, m, rows = 3, 3, 256 * 40 + 77
    agg = ctx.aggregate(n, m)
    ref = orc.State(orc.FAITHFUL)

    def batch(make_keys):
        num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
        cat = [make_keys().astype(np.int32) for _ in range(m)]
        dn = [torch.from_numpy(c).cuda() for c in num]
        dc = [torch.from_numpy(c).cuda() for c in cat]
        torch.cuda.synchronize()
        agg.update_device(dn, dc)
        ref.update(num, cat)
        assert blob_to_dict(agg.finalize()) == blob_to_dict(ref.finalize())

    batch(lambda: rng.integers(0, 6, rows))                       # first update: dictionary pass
    batch(lambda: rng.integers(0, 6, rows))                       # all keys known: nothing skipped

    def few_new():
        k = rng.integers(0, 6, rows)
        k[256 * 7 + 3] = 100                                      # one new key in tile 7
        k[256 * 31 + 200] = -5                                    # another in tile 31
        k[rows - 1] = 77                                          # and one in the < 256-row tail
        return k
    batch(few_new)
    batch(lambda: rng.integers(0, 14, rows))                      # new keys in every tile, <= 16 keys total? (6+3+...)
    batch(lambda: rng.integers(0, 40, rows))                      # outgrows the fused kernel
    batch(lambda: rng.integers(0, 40, rows))
    agg.close()


def test_mixed_random_floats_10_10(ctx):
    rng = np.random.default_rng(77)
    rows, n, m = 400_000, 10, 10
    num = [rng.random(rows, dtype=np.float32) for _ in range(n)]
    cat = [rng.integers(0, 16, rows).astype(np.int32) for _ in range(m)]
    got = blob_to_dict(gpu_triple(ctx, num, cat))
    want = blob_to_dict(orc.State(orc.WIDE).update(num, cat, threads=8).finalize())
    assert_triple_close(got, want, rtol=RTOL)            # counts / keys exact inside


def test_random_shapes_exact(ctx):
    """Seeded sweep over shapes that land on every kernel combination (dense only, fused, fused +
    tail, multi-pass LDS tables, HBM tables, NB kind): integer-valued tables, exact equality."""
    import torch
    rng = np.random.default_rng(2024)
    for trial in range(48):
        n = int(rng.integers(0, 21))
        m = int(rng.integers(0, 21))
        if n == 0 and m == 0:
            n = 1
        nb = bool(rng.integers(0, 4) == 0)
        keys = int(rng.choice([2, 5, 16, 17, 40, 300]))
        rows = int(rng.choice([1, 255, 256, 257, 1000, 4096, 10_007, 33_333]))
        num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
        cat = [(rng.integers(0, keys, rows) * 3 - 7).astype(np.int32) for _ in range(m)]
        agg = ctx.aggregate(n, m, cofactor_hip.NB if nb else cofactor_hip.TRIPLE)
        ref = orc.State(orc.FAITHFUL)
        for part in range(int(rng.integers(1, 3))):            # one or two updates into the same state
            lo = 0 if part == 0 else rows // 3
            dn = [torch.from_numpy(np.ascontiguousarray(c[lo:])).cuda() for c in num]
            dc = [torch.from_numpy(np.ascontiguousarray(c[lo:])).cuda() for c in cat]
            torch.cuda.synchronize()
            agg.update_device(dn, dc)
            ref.update([c[lo:] for c in num], [c[lo:] for c in cat], nb=nb)
        got, want = blob_to_dict(agg.finalize()), blob_to_dict(ref.finalize())
        agg.close()
        assert got == want, (trial, n, m, nb, keys, rows)


@pytest.mark.parametrize("n,m,keys", [(20, 0, 0), (3, 2, 6), (10, 10, 16), (0, 3, 40), (5, 4, 300), (13, 0, 0), (16, 0, 0), (15, 2, 5)])
def test_masked_update_equals_filtered_rows(ctx, n, m, keys):
    """cofactor_agg_update_device_masked == the aggregate over the kept rows only (the WHERE
    <col>_IS_NULL IS FALSE filter of the MICE drivers), N included; mixed with an unmasked update."""
    import torch
    rng = np.random.default_rng(500 + n + m)
    rows = 30_011
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    cat = [rng.integers(-1, max(1, keys) - 1, rows).astype(np.int32) for _ in range(m)]
    mask = (rng.random(rows) < 0.9).astype(np.uint8)
    mask[:300] = 0                                            # a whole tile filtered out
    dn = [torch.from_numpy(c).cuda() for c in num]
    dc = [torch.from_numpy(c).cuda() for c in cat]
    dm = torch.from_numpy(mask).cuda()
    torch.cuda.synchronize()
    agg = ctx.aggregate(n, m)
    agg.update_device_masked(dn, dc, dm)
    keep = mask.astype(bool)
    ref = orc.State(orc.FAITHFUL).update([c[keep] for c in num], [c[keep] for c in cat])
    assert blob_to_dict(agg.finalize()) == blob_to_dict(ref.finalize())
    agg.update_device(dn, dc)                                 # then everything, unmasked
    ref.update(num, cat)
    assert blob_to_dict(agg.finalize()) == blob_to_dict(ref.finalize())
    agg.reset()
    assert blob_to_dict(agg.finalize())["N"] == 0
    agg.close()


def test_masked_update_after_reset_skips_the_dictionary_pass_and_still_meets_new_keys(ctx):
    """Second and later masked updates (each column of a MICE sweep) run without the dictionary
    pass; tiles in which a key shows up that the dictionaries do not hold yet — in a kept or in a
    filtered row — are redone with their slice of the filter."""
    import torch
    rng = np.random.default_rng(77)
    rows, n, m = 40_000, 3, 4
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    cat = [rng.integers(0, 5, rows).astype(np.int32) for _ in range(m)]
    mask = (rng.random(rows) < 0.9).astype(np.uint8)
    agg = ctx.aggregate(n, m)
    dev = lambda cols: [torch.from_numpy(np.ascontiguousarray(c)).cuda() for c in cols]
    agg.update_device_masked(dev(num), dev(cat), torch.from_numpy(mask).cuda())   # builds the dictionaries
    agg.reset()
    cat2 = [c.copy() for c in cat]
    cat2[1][1000] = 91; mask[1000] = 1             # new key in a kept row
    cat2[2][20_000] = 92; mask[20_000] = 0         # new key in a filtered row
    cat2[0][39_990] = 93; mask[39_990] = 1         # new key in the < 256-row tail
    dm = torch.from_numpy(mask).cuda()
    ctx.profile(True); ctx.profile_read()
    agg.update_device_masked(dev(num), dev(cat2), dm)
    keep = mask.astype(bool)
    ref = orc.State(orc.FAITHFUL).update([c[keep] for c in num], [c[keep] for c in cat2])
    assert blob_to_dict(agg.finalize()) == blob_to_dict(ref.finalize())
    prof = ctx.profile_read(); ctx.profile(False)
    assert prof["fused_launches"] >= 1
    agg.close()


@pytest.mark.parametrize("shift", [1, 2, 3])
@pytest.mark.parametrize("n,m", [(5, 0), (3, 2)])
def test_masked_update_with_a_filter_that_is_not_word_aligned(ctx, n, m, shift):
    """The kernels read the row filter of whole tiles as 4-byte words; a filter starting 1..3 bytes
    off a word boundary (a slice of a larger buffer) must give the same triple."""
    import torch
    rng = np.random.default_rng(600 + shift)
    rows = 5_000 + shift
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    cat = [rng.integers(0, 6, rows).astype(np.int32) for _ in range(m)]
    mask = (rng.random(rows) < 0.8).astype(np.uint8)
    big = torch.zeros(rows + 8, dtype=torch.uint8, device="cuda")
    dm = big[shift:shift + rows]
    dm.copy_(torch.from_numpy(mask))
    assert dm.data_ptr() % 4 == shift
    dn = [torch.from_numpy(c).cuda() for c in num]
    dc = [torch.from_numpy(c).cuda() for c in cat]
    torch.cuda.synchronize()
    agg = ctx.aggregate(n, m)
    agg.update_device_masked(dn, dc, dm)
    keep = mask.astype(bool)
    ref = orc.State(orc.FAITHFUL).update([c[keep] for c in num], [c[keep] for c in cat])
    assert blob_to_dict(agg.finalize()) == blob_to_dict(ref.finalize())
    agg.close()


def test_fused_per_key_sums_with_a_single_key_stay_far_inside_the_tolerance(ctx):
    """The fused kernel keeps per-key partial sums in fp32 MFMA accumulators between fp64 folds;
    the longest fp32 chains arise when a column has ONE key (every row lands in the same cell).
    Heavy-tailed values, 4 M rows: the sums must still be within 1e-6 of the fp64 sums, an order of
    magnitude inside the 1e-5 the path promises."""
    import torch
    rows, n, m = 4_000_000, 10, 10
    g = torch.Generator(device="cuda").manual_seed(3)
    num = [torch.exp(3 * torch.randn(rows, generator=g, device="cuda")).contiguous() for _ in range(n)]
    cat = [torch.full((rows,), 7 + c, dtype=torch.int32, device="cuda") for c in range(m)]
    agg = ctx.aggregate(n, m)
    ctx.profile(True); ctx.profile_read()
    agg.update_device(num, cat)
    t = blob_to_dict(agg.finalize())
    prof = ctx.profile_read(); ctx.profile(False)
    agg.close()
    assert prof["fused_launches"] >= 1
    for k in range(n):
        want = float(num[k].double().sum())
        for c in (0, 4, 9):
            (entry,) = t["quad_num_cat"][k * m + c]
            assert entry["key"] == 7 + c
            assert abs(entry["value"] - want) <= 1e-6 * want, (k, c)


def test_nb_aggregate(ctx):
    rng = np.random.default_rng(21)
    rows = 30_000
    num, cat = int_table(rng, rows, 6, 4)
    got = blob_to_dict(gpu_triple(ctx, num, cat, nb=True))
    want = blob_to_dict(orc.State(orc.FAITHFUL).update(num, cat, nb=True).finalize())
    assert got == want and "quad_cat" not in got and len(got["quad_agg"]) == 6


# ---- life cycle: combine, reset, empty, host chunks with selection vectors --------------------
def test_combine_of_shards_equals_whole(ctx):
    rng = np.random.default_rng(31)
    rows = 30_000
    num, cat = int_table(rng, rows, 4, 3)
    cut = 12_345
    a, b = ctx.aggregate(4, 3), ctx.aggregate(4, 3)
    a.update_host([c[:cut] for c in num], [c[:cut] for c in cat])
    b.update_host([c[cut:] for c in num], [c[cut:] for c in cat])
    a.combine(b)
    want = blob_to_dict(orc.State(orc.FAITHFUL).update(num, cat).finalize())
    assert blob_to_dict(a.finalize()) == want
    # src is unchanged by combine
    assert blob_to_dict(b.finalize())["N"] == rows - cut
    a.close(); b.close()


def test_empty_and_reset(ctx):
    agg = ctx.aggregate(3, 2)
    d = blob_to_dict(agg.finalize())
    assert d["N"] == 0 and d["lin_agg"] == [0.0] * 3 and d["lin_cat"] == [[], []]
    rng = np.random.default_rng(1)
    num, cat = int_table(rng, 1000, 3, 2)
    agg.update_host(num, cat)
    assert blob_to_dict(agg.finalize())["N"] == 1000
    agg.reset()
    assert blob_to_dict(agg.finalize()) == d
    agg.update_host(num, cat)                              # dictionaries survive a reset
    assert blob_to_dict(agg.finalize()) == blob_to_dict(orc.State(orc.FAITHFUL).update(num, cat).finalize())
    agg.close()


def test_host_chunks_with_selection_vectors_group_by(ctx):
    """What DuckDB's update delivers: 2048-row chunks, per-column selection vectors
    (dictionary / filtered vectors) and per-row state pointers (GROUP BY)."""
    rng = np.random.default_rng(8)
    rows, n, m, groups = 10_000, 3, 2, 4
    num, cat = int_table(rng, rows, n, m)
    gid = rng.integers(0, groups, rows).astype(np.int32)
    aggs = [ctx.aggregate(n, m) for _ in range(groups)]
    for lo in range(0, rows, 2048):
        hi = min(rows, lo + 2048)
        cnt = hi - lo
        # physical columns hold the chunk's values permuted; sel undoes the permutation
        perm = rng.permutation(cnt).astype(np.uint32)
        inv = np.argsort(perm).astype(np.uint32)
        pnum = [c[lo:hi][perm] for c in num]
        pcat = [c[lo:hi][perm] for c in cat]
        for g in range(groups):
            idx = np.nonzero(gid[lo:hi] == g)[0].astype(np.uint32)
            if len(idx):
                aggs[g].update_host(pnum, pcat, num_sel=[inv] * n, cat_sel=[inv] * m, row_idx=idx)
    want = orc.grouped_update(num, cat, gid, groups, mode=orc.FAITHFUL)
    for g in range(groups):
        assert blob_to_dict(aggs[g].finalize()) == blob_to_dict(want[g].finalize())
        aggs[g].close()


def test_group_by_with_many_states_and_growing_staging(ctx):
    """A GROUP BY with hundreds of states: each state's staging starts at 512 rows and grows only
    with the rows it receives (one big group crosses several growth steps, the rest stay small)."""
    rng = np.random.default_rng(99)
    rows, n, m, groups = 120_000, 2, 2, 300
    num, cat = int_table(rng, rows, n, m)
    gid = rng.integers(1, groups, rows).astype(np.int32)
    gid[rng.random(rows) < 0.7] = 0                          # group 0 takes most rows
    aggs = [ctx.aggregate(n, m) for _ in range(groups)]
    for lo in range(0, rows, 2048):
        hi = min(rows, lo + 2048)
        order = np.argsort(gid[lo:hi], kind="stable").astype(np.uint32)
        bounds = np.searchsorted(gid[lo:hi][order], np.arange(groups + 1))
        pnum, pcat = [c[lo:hi] for c in num], [c[lo:hi] for c in cat]
        for g in np.unique(gid[lo:hi]):
            aggs[g].update_host(pnum, pcat, row_idx=order[bounds[g]:bounds[g + 1]])
    want = orc.grouped_update(num, cat, gid, groups, mode=orc.FAITHFUL)
    for g in range(groups):
        assert blob_to_dict(aggs[g].finalize()) == blob_to_dict(want[g].finalize()), g
        aggs[g].close()


def test_concurrent_states_on_one_context(ctx):
    """DuckDB calls update from several worker threads at once, each on its own states; all of them
    share one cofactor_ctx (stream + scratch buffers).  ctypes releases the GIL, so these really
    overlap inside the library."""
    import threading
    rng = np.random.default_rng(123)
    n, m, rows, workers = 6, 3, 40_000, 6
    tables = [int_table(np.random.default_rng(1000 + w), rows, n, m, k=5 + 3 * w) for w in range(workers)]
    aggs = [ctx.aggregate(n, m) for _ in range(workers)]
    errors = []

    def work(w):
        try:
            num, cat = tables[w]
            for lo in range(0, rows, 2048):                  # DataChunk-sized host updates
                hi = min(rows, lo + 2048)
                aggs[w].update_host([c[lo:hi] for c in num], [c[lo:hi] for c in cat])
            aggs[w].finalize()
        except Exception as e:                                 # noqa: BLE001
            errors.append((w, e))

    threads = [threading.Thread(target=work, args=(w,)) for w in range(workers)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for w in range(workers):
        want = blob_to_dict(orc.State(orc.FAITHFUL).update(*tables[w]).finalize())
        assert blob_to_dict(aggs[w].finalize()) == want, w
        aggs[w].close()


def test_fused_equals_unfused_large(ctx):
    """sum_to_triple == sum_triple(to_cofactor(.)) (test_sum.py:40-52) beyond the 5-row table."""
    rng = np.random.default_rng(55)
    num, cat = int_table(rng, 500, 3, 2)
    fused = blob_to_dict(gpu_triple(ctx, num, cat))
    agg = ctx.aggregate(3, 2)
    agg.update_triples(cofactor_hip.lift_host(num, cat))
    assert blob_to_dict(agg.finalize()) == fused
    agg.close()


@pytest.mark.parametrize("n,m,nb", [(5, 2, False), (20, 0, False), (1, 0, False), (12, 3, True), (7, 0, True)])
def test_dense_export_import_roundtrip(ctx, n, m, nb):
    """The dense seam on one GPU: export kernel -> (x2, standing in for a 2-rank all-reduce of
    equal shards) -> import kernel doubles N, lin and quad and leaves the categorical part alone.
    Half of the rows come in through a host-side merge (combine), so the export also carries the
    addends the state holds on the host."""
    import torch
    rng = np.random.default_rng(66)
    num, cat = int_table(rng, 9000, n, m)
    kind = cofactor_hip.NB if nb else cofactor_hip.TRIPLE
    agg, other = ctx.aggregate(n, m, kind), ctx.aggregate(n, m, kind)
    agg.update_host([c[:4000] for c in num], [c[:4000] for c in cat])
    other.update_host([c[4000:] for c in num], [c[4000:] for c in cat])
    agg.combine(other)
    other.close()
    before = blob_to_dict(agg.finalize())
    assert before == blob_to_dict(orc.State(orc.WIDE).update(num, cat, nb=nb).finalize())
    buf = torch.zeros(agg.dense_len(), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()                      # torch's fill must not race the library's stream
    agg.export_dense_device(buf.data_ptr())
    ctx.synchronize()
    host = buf.cpu().numpy()
    assert host[0] == 9000 and list(host[1:1 + n]) == before["lin_agg"]
    assert list(host[1 + n:]) == before["quad_agg"]
    buf *= 2
    torch.cuda.synchronize()
    agg.import_dense_device(buf.data_ptr())
    after = blob_to_dict(agg.finalize())
    assert after["N"] == 18000 and after["quad_agg"] == [2 * v for v in before["quad_agg"]]
    assert after["lin_agg"] == [2 * v for v in before["lin_agg"]]
    assert after["lin_cat"] == before["lin_cat"]
    if not nb:
        assert after["quad_cat"] == before["quad_cat"] and after["quad_num_cat"] == before["quad_num_cat"]
    agg.close()


@pytest.mark.parametrize("n,m,nb,keys", [(3, 2, False, 6), (10, 10, False, 16), (2, 3, False, 70), (0, 2, False, 5),
                                          (4, 2, True, 9)])
def test_table_seam_alignment_and_roundtrip(ctx, n, m, nb, keys):
    """The categorical seam on one GPU: a state is aligned to a global key list that holds keys it
    never saw (another rank's), keeps its triple, and its table image exported -> x3 -> imported
    triples every count and per-key sum.  Part of the rows sit on the host side of the state
    (merged in by combine), so the alignment also folds host-held keys into the tables."""
    import torch
    rng = np.random.default_rng(67)
    rows = 20_000
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    cat = [rng.integers(-2, keys - 2, rows).astype(np.int32) for _ in range(m)]
    kind = cofactor_hip.NB if nb else cofactor_hip.TRIPLE
    agg, other = ctx.aggregate(n, m, kind), ctx.aggregate(n, m, kind)
    keep = []                                     # updates are asynchronous: the columns must outlive them

    def d(cols, a, b):
        keep.append([torch.from_numpy(c[a:b]).cuda() for c in cols])
        return keep[-1]
    agg.update_device(d(num, 0, 15_000), d(cat, 0, 15_000))
    other.update_device(d(num, 15_000, rows), d(cat, 15_000, rows))
    agg.combine(other)                            # on the device: both states aligned to the union of their keys
    other.close()
    assert agg.dict_signature() != 0
    # a few lifted rows on top (sum_triple's blob form): they sit on the HOST side of the state
    extra_num = [c[:7].copy() for c in num]
    extra_cat = [c[:7].copy() for c in cat]
    agg.update_triples(cofactor_hip.lift_host(extra_num, extra_cat, kind))
    num = [np.concatenate([c, e]) for c, e in zip(num, extra_num)]
    cat = [np.concatenate([c, e]) for c, e in zip(cat, extra_cat)]
    want = blob_to_dict(orc.State(orc.WIDE).update(num, cat, nb=nb).finalize())
    assert agg.dict_signature() == 0
    own, offs = agg.keys()
    assert [list(own[int(offs[c]):int(offs[c + 1])]) for c in range(m)] == \
           [sorted(set(c.tolist())) for c in cat]
    # the "other ranks'" keys: strangers on both sides of every column's range, duplicates, unsorted
    glob, goffs = [], [0]
    for c in range(m):
        mine = own[int(offs[c]):int(offs[c + 1])]
        lst = np.concatenate([mine[::-1], [1000 + c, -1000 - c, 1000 + c], mine[:3]]).astype(np.int32)
        glob.append(lst)
        goffs.append(goffs[-1] + lst.size)
    agg.align_keys(np.concatenate(glob), np.array(goffs, dtype=np.uint64))
    sig = agg.dict_signature()
    assert sig != 0
    assert blob_to_dict(agg.finalize()) == want          # zero-count keys do not show
    tlen = int(agg.tables_len())
    assert tlen > 0
    buf = torch.zeros(tlen, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    agg.export_tables_device(buf.data_ptr())
    ctx.synchronize()
    buf *= 3
    torch.cuda.synchronize()
    agg.import_tables_device(buf.data_ptr())
    got = blob_to_dict(agg.finalize())
    tripled = lambda lists: [[dict(e, value=3 * e["value"]) for e in l] for l in lists]
    assert got["lin_cat"] == tripled(want["lin_cat"])
    if not nb:
        assert got["quad_num_cat"] == tripled(want["quad_num_cat"])
        assert got["quad_cat"] == tripled(want["quad_cat"])
    assert got["lin_agg"] == want["lin_agg"] and got["N"] == want["N"]
    # more rows with known keys keep the alignment; a new key ends it
    agg.update_device(d(num, 0, 3000), d(cat, 0, 3000))
    assert agg.dict_signature() == sig
    fresh = [torch.full((300,), 7777, dtype=torch.int32, device="cuda") for _ in range(m)]
    ones = [torch.ones(300, device="cuda") for _ in range(n)]
    agg.update_device(ones, fresh)
    assert agg.dict_signature() == 0
    ctx.synchronize()
    agg.close()


@pytest.mark.parametrize("pref", ["1", "2", "3", "4"])
@pytest.mark.parametrize("n,m,nb,keys", [(10, 10, False, 16), (3, 2, False, 7), (0, 3, False, 16), (5, 1, False, 3),
                                          (20, 4, False, 9), (12, 7, False, 16), (16, 6, False, 12), (4, 3, True, 16),
                                          (0, 2, True, 5), (20, 10, True, 16), (7, 9, False, 2)])
def test_both_one_pass_kernels_exact(monkeypatch, pref, n, m, nb, keys):
    """COFACTOR_FUSED=1 pins fused_kernel (three teams), =2 fused2_kernel (LDS-DMA ring, one-hot
    MFMAs), =3 fused3_kernel (specialised waves), =4 nb_ring_kernel (NB kind only); shapes a kernel
    does not take go through the others / the two-kernel path.  Integer-valued table,
    ragged row count, keys below zero (hash probe instead of the byte table) in every other column,
    plain and filtered updates, then a second update in optimistic mode that brings a new key:
    every count and sum must equal the oracle's exactly."""
    import torch
    monkeypatch.setenv("COFACTOR_FUSED", pref)
    c2 = cofactor_hip.Context(0)
    rng = np.random.default_rng(1000 + 7 * n + m)
    rows = 70_000 + 37
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    cat = [(rng.integers(0, keys, rows) - (3 if c % 2 else 0)).astype(np.int32) for c in range(m)]
    mask = (rng.random(rows) < 0.6).astype(np.uint8)
    kind = cofactor_hip.NB if nb else cofactor_hip.TRIPLE
    dn = [torch.from_numpy(c).cuda() for c in num]
    dc = [torch.from_numpy(c).cuda() for c in cat]
    dm = torch.from_numpy(mask).cuda()
    torch.cuda.synchronize()
    for use_mask in (False, True):
        agg = c2.aggregate(n, m, kind)
        if use_mask:
            agg.update_device_masked(dn, dc, dm)
            sel = mask.astype(bool)
        else:
            agg.update_device(dn, dc)
            sel = np.ones(rows, dtype=bool)
        want = orc.State(orc.WIDE).update([c[sel] for c in num], [c[sel] for c in cat], nb=nb)
        assert blob_to_dict(agg.finalize()) == blob_to_dict(want.finalize()), use_mask
        # second batch, dictionaries known (optimistic pass), one new key in the middle of it
        cat2 = [c.copy() for c in cat]
        if keys < 16:
            cat2[0][40_000:40_003] = 12345
        dc2 = [torch.from_numpy(c).cuda() for c in cat2]
        torch.cuda.synchronize()
        if use_mask:
            agg.update_device_masked(dn, dc2, dm)
        else:
            agg.update_device(dn, dc2)
        want.update([c[sel] for c in num], [c[sel] for c in cat2], nb=nb)
        assert blob_to_dict(agg.finalize()) == blob_to_dict(want.finalize()), ("second", use_mask)
        agg.close()
    c2.close()


@pytest.mark.parametrize("n,m", [(20, 20), (20, 10), (13, 11), (3, 12), (10, 20)])
@pytest.mark.parametrize("masked", [False, True])
def test_wide_shapes_take_their_per_key_sums_through_sub_launches(monkeypatch, n, m, masked):
    """Shapes beyond one one-pass launch (m > 10, or too many per-key-sum blocks) with <= 16 keys per
    column: key counts and per-key sums come from fused2_kernel sub-launches over column groups
    (whole 256-row tiles) plus the LDS-atomic kernel for the tail; pairs from the code cache.
    Exact against the faithful oracle on integer data, with and without a row filter, in two
    batches (the second brings new keys), and equal to the run with the sub-launches switched off."""
    import torch
    rng = np.random.default_rng(100 * n + m + masked)
    blobs = []
    for no_sub in ("0", "1"):
        monkeypatch.setenv("COFACTOR_NO_SUB", no_sub)
        c = cofactor_hip.Context(0)
        agg = c.aggregate(n, m)
        ref = orc.State(orc.FAITHFUL)
        r = np.random.default_rng(100 * n + m + masked)
        for rows, k in ((70_001, 9), (33_333, 16)):
            num = [r.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
            cat = [(r.integers(0, k, rows) * 5 - 20).astype(np.int32) for _ in range(m)]
            dn = [torch.from_numpy(x).cuda() for x in num]
            dc = [torch.from_numpy(x).cuda() for x in cat]
            if masked:
                keep = (r.random(rows) < 0.7).astype(np.uint8)
                torch.cuda.synchronize()
                agg.update_device_masked(dn, dc, torch.from_numpy(keep).cuda())
                sel = keep.astype(bool)
                ref.update([x[sel] for x in num], [x[sel] for x in cat])
            else:
                torch.cuda.synchronize()
                agg.update_device(dn, dc)
                ref.update(num, cat)
        blob = agg.finalize()
        agg.close(); c.close()
        assert blob_to_dict(blob) == blob_to_dict(ref.finalize())
        blobs.append(blob)
    assert np.array_equal(blobs[0], blobs[1])


def test_staging_blocks_are_reused_by_the_next_state(ctx):
    """The DataChunk path's pinned staging blocks go to a pool in the context when a state dies or
    outgrows them; the next state of the same shape takes them over.  Three generations of states
    over the same host table (enough rows to grow the block twice) must all give the oracle's triple."""
    rng = np.random.default_rng(77)
    rows = 150_000
    num, cat = int_table(rng, rows, 4, 2)
    want = blob_to_dict(orc.State(orc.FAITHFUL).update(num, cat).finalize())
    for gen in range(3):
        aggs = [ctx.aggregate(4, 2) for _ in range(3)]
        for i, a in enumerate(aggs):
            lo, hi = rows * i // 3, rows * (i + 1) // 3
            for c0 in range(lo, hi, 2048):                       # DuckDB-sized chunks
                c1 = min(hi, c0 + 2048)
                a.update_host([c[c0:c1] for c in num], [c[c0:c1] for c in cat])
        aggs[0].combine(aggs[1]); aggs[0].combine(aggs[2])
        got = blob_to_dict(aggs[0].finalize())
        for a in aggs:
            a.close()
        assert got == want, gen


@pytest.mark.parametrize("n,m,k1,k2", [(2, 12, 6, 9), (3, 4, 40, 70), (2, 12, 10, 30), (0, 11, 5, 7)])
def test_code_cache_route_without_a_dictionary_pass_meets_new_keys(ctx, n, m, k1, k2):
    """Shapes no one-pass kernel takes: from the second batch on the dictionary pass is skipped and the
    code translation reports keys it does not know before anything is accumulated.  Batch 2 repeats
    batch 1's keys (no pass), batch 3 brings new keys (miss -> pass -> redo), batch 4 is filtered and
    brings one more; reset keeps the dictionaries.  All against the oracle, exactly."""
    import torch
    rng = np.random.default_rng(9000 + 10 * n + m + k1)
    agg = ctx.aggregate(n, m)
    ref = orc.State(orc.FAITHFUL)
    for step, (rows, k) in enumerate(((30_011, k1), (41_000, k1), (25_003, k2), (33_333, k2 + 1))):
        num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
        cat = [(rng.integers(0, k, rows) * 3 - 7).astype(np.int32) for _ in range(m)]
        dn = [torch.from_numpy(x).cuda() for x in num]
        dc = [torch.from_numpy(x).cuda() for x in cat]
        torch.cuda.synchronize()
        if step == 3:
            keep = (rng.random(rows) < 0.5).astype(np.uint8)
            keep[:8] = 1
            agg.update_device_masked(dn, dc, torch.from_numpy(keep).cuda())
            sel = keep.astype(bool)
            ref.update([x[sel] for x in num], [x[sel] for x in cat])
        else:
            agg.update_device(dn, dc)
            ref.update(num, cat)
        assert blob_to_dict(agg.finalize()) == blob_to_dict(ref.finalize()), step
    agg.reset()
    agg.update_device(dn, dc)
    assert blob_to_dict(agg.finalize()) == blob_to_dict(orc.State(orc.FAITHFUL).update(num, cat).finalize())
    agg.close()


@pytest.mark.parametrize("n,m,keys", [(10, 10, 16), (3, 2, 300), (0, 4, 1000), (20, 1, 40), (7, 20, 9)])
def test_nb_ring_kernel_exact_at_any_cardinality(n, m, keys):
    """nb_ring_kernel (default for sum_to_nb_agg): one row per lane, counts by LDS atomics, any
    cardinality whose tables fit LDS — 16-bit code tables, keys below zero through the hash probe,
    row filter, a second batch in optimistic mode with a new key.  Exact against the oracle."""
    import torch
    c2 = cofactor_hip.Context(0)
    rng = np.random.default_rng(4000 + 11 * n + m)
    rows = 150_000 + 41
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    cat = [(rng.integers(0, keys, rows) - (keys // 3 if c % 2 else 0)).astype(np.int32) for c in range(m)]
    mask = (rng.random(rows) < 0.7).astype(np.uint8)
    dn = [torch.from_numpy(c).cuda() for c in num]
    dc = [torch.from_numpy(c).cuda() for c in cat]
    dm = torch.from_numpy(mask).cuda()
    torch.cuda.synchronize()
    for use_mask in (False, True):
        agg = c2.aggregate(n, m, cofactor_hip.NB)
        sel = mask.astype(bool) if use_mask else np.ones(rows, dtype=bool)
        (agg.update_device_masked(dn, dc, dm) if use_mask else agg.update_device(dn, dc))
        want = orc.State(orc.WIDE).update([c[sel] for c in num], [c[sel] for c in cat], nb=True)
        assert blob_to_dict(agg.finalize()) == blob_to_dict(want.finalize()), use_mask
        cat2 = [c.copy() for c in cat]
        cat2[0][70_000:70_005] = 123456                       # a key no dictionary holds yet
        dc2 = [torch.from_numpy(c).cuda() for c in cat2]
        torch.cuda.synchronize()
        (agg.update_device_masked(dn, dc2, dm) if use_mask else agg.update_device(dn, dc2))
        want.update([c[sel] for c in num], [c[sel] for c in cat2], nb=True)
        assert blob_to_dict(agg.finalize()) == blob_to_dict(want.finalize()), ("second", use_mask)
        c2.synchronize()
        agg.close()
    c2.close()


@pytest.mark.parametrize("n,keys", [(10, (64,) * 10), (3, (17, 40, 64, 5)), (10, (33, 20, 64, 48, 7, 64, 30, 25, 61, 18, 64, 50)),
                                    (1, (64, 64)), (5, (32, 31, 17))])
def test_per_key_sums_of_17_to_64_keys_on_the_matrix_cores(ctx, n, keys):
    """cat_sums_mfma_kernel (catsums.hip): key columns of 17 .. 64 keys — counts and per-key sums as
    one-hot x bf16-piece products.  Exact on integer tables (also a row count that ends inside a tile,
    two batches, a filtered batch), within 1e-5 of the wide oracle on floats, inf / nan only where the
    oracle has them."""
    import torch
    rng = np.random.default_rng(sum(keys) + n)
    rows, m = 40_037, len(keys)
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    cat = [(rng.integers(0, k, rows) * 3 - 17).astype(np.int32) for k in keys]
    got = blob_to_dict(gpu_triple(ctx, num, cat))
    want = blob_to_dict(orc.State(orc.FAITHFUL).update(num, cat).finalize())
    assert got == want
    # two batches, the second filtered
    keep = (rng.random(rows) < 0.7).astype(np.uint8)
    agg = ctx.aggregate(n, m)
    dn = [torch.from_numpy(c).cuda() for c in num]
    dc = [torch.from_numpy(c).cuda() for c in cat]
    dk = torch.from_numpy(keep).cuda()
    torch.cuda.synchronize()
    agg.update_device(dn, dc)
    agg.update_device_masked(dn, dc, dk)
    got2 = blob_to_dict(agg.finalize())
    agg.close()
    sel = keep.astype(bool)
    ref = orc.State(orc.FAITHFUL).update(num, cat).update([c[sel] for c in num], [c[sel] for c in cat])
    assert got2 == blob_to_dict(ref.finalize())
    # floats, and a non-finite value
    fnum = [(rng.random(rows) * 200 - 100).astype(np.float32) for _ in range(n)]
    fnum[0][123] = np.inf
    g = blob_to_dict(gpu_triple(ctx, fnum, cat))
    w = blob_to_dict(orc.State(orc.WIDE).update(fnum, cat).finalize())
    assert g["lin_cat"] == w["lin_cat"] and g["quad_cat"] == w["quad_cat"]
    for gl, wl in zip(g["quad_num_cat"], w["quad_num_cat"]):
        assert [e["key"] for e in gl] == [e["key"] for e in wl]
        gv, wv = np.array([e["value"] for e in gl], dtype=np.float64), np.array([e["value"] for e in wl], dtype=np.float64)
        assert np.array_equal(np.isfinite(gv), np.isfinite(wv))
        fin = np.isfinite(wv)
        assert np.allclose(gv[fin], wv[fin], rtol=1e-5, atol=1e-3)
