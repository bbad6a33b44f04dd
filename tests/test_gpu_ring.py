"""Batched ring operations on the GPU against the oracle and the reference's
golden vectors: to_cofactor (lift kernel), sum_triple (column-reduce + key-list kernels),
multiply_triple (block-assembly kernels), their NB variants, the GROUP BY state pool, and the
reference README's factorised-join query end to end through the C ABI."""
import numpy as np
import pytest

import cofactor_hip
from cofactor_hip import ring
from golden_cases import cases
from oracle import oracle as orc
from triple_fmt import blob_to_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = cofactor_hip.Context(0)
    yield c
    c.close()


def _cuda(cols):
    import torch
    out = [torch.from_numpy(np.ascontiguousarray(c)).cuda() for c in cols]
    torch.cuda.synchronize()
    return out


class RingBackend:
    """The golden cases' backend with every op on the GPU kernels."""

    def __init__(self, ctx, where):
        self.ctx, self.where = ctx, where            # where: "device" | "host" entry points

    def sum_to(self, num, cat, nb):
        agg = self.ctx.aggregate(len(num), len(cat), cofactor_hip.NB if nb else cofactor_hip.TRIPLE)
        agg.update_host(num, cat)
        b = agg.finalize()
        agg.close()
        return b

    def lift(self, num, cat, nb):
        kind = cofactor_hip.NB if nb else cofactor_hip.TRIPLE
        num = [np.ascontiguousarray(c, dtype=np.float32) for c in num]
        cat = [np.ascontiguousarray(c, dtype=np.int32) for c in cat]
        if self.where == "device":
            return ring.lift_device(self.ctx, _cuda(num), _cuda(cat), kind).to_blobs()
        return ring.lift_host(self.ctx, num, cat, kind).to_blobs()

    def sum_lifted(self, blobs, nb):
        n, m = int(blobs[0][1]), int(blobs[0][2])
        agg = self.ctx.aggregate(n, m, cofactor_hip.NB if nb else cofactor_hip.TRIPLE)
        ring.update_tvec(agg, ring.tvec_from_blobs(blobs, device="cuda" if self.where == "device" else None))
        b = agg.finalize()
        agg.close()
        return b

    def multiply(self, a, b, nb):
        dev = "cuda" if self.where == "device" else None
        out = ring.multiply(self.ctx, ring.tvec_from_blobs([a], device=dev), ring.tvec_from_blobs([b], device=dev))
        return out.to_blobs()[0]


def _load():
    import json
    import os
    root = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(root, "golden", "ring_goldens.json")) as fh:
        g = json.load(fh)
    from conftest import RefTable
    return g, {f: RefTable(v["table"]) for f, v in g.items()}


_G, _T = _load()
_CASES = cases(_G, _T)


@pytest.mark.parametrize("where", ["device", "host"])
@pytest.mark.parametrize("case", _CASES, ids=[c[0] for c in _CASES])
def test_reference_goldens_through_the_ring_kernels(ctx, case, where):
    """test_lift.py / test_mul.py / test_sum.py (+ NB) literals with lift, sum_triple and multiply
    all executed by the HIP kernels, through the device and through the host entry points."""
    pairs = case[1](RingBackend(ctx, where))
    assert pairs
    for got, want in pairs:
        assert got == want


@pytest.mark.parametrize("n,m,nb", [(3, 2, False), (0, 3, False), (4, 0, False), (20, 20, False), (5, 4, True), (0, 2, True),
                                     (7, 0, True)])
def test_lift_kernel_equals_oracle(ctx, n, m, nb):
    rng = np.random.default_rng(300 + n + m)
    rows = 3001
    num = [rng.normal(size=rows).astype(np.float32) for _ in range(n)]
    cat = [rng.integers(-5, 40, rows).astype(np.int32) for _ in range(m)]
    got = ring.lift_device(ctx, _cuda(num), _cuda(cat), cofactor_hip.NB if nb else cofactor_hip.TRIPLE).to_blobs()
    want = orc.lift(num, cat, nb=nb)
    assert len(got) == rows
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)          # float products: bit-identical


@pytest.mark.parametrize("n,m,nb", [(3, 2, False), (10, 10, False), (20, 0, False), (0, 4, False), (6, 3, True)])
def test_sum_triple_of_lifted_rows_equals_the_fused_aggregate(ctx, n, m, nb):
    """sum_triple(to_cofactor(cols)) == sum_to_triple(cols) (test_sum.py:40-52) at 300 000 rows, all on
    the device: lift kernel -> column-reduce / key-list kernels against the one-pass aggregate."""
    rng = np.random.default_rng(400 + n + m)
    rows = 300_000
    kind = cofactor_hip.NB if nb else cofactor_hip.TRIPLE
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    cat = [rng.integers(-3, 9, rows).astype(np.int32) for _ in range(m)]
    dn, dc = _cuda(num), _cuda(cat)
    lifted = ring.lift_device(ctx, dn, dc, kind)
    a = ctx.aggregate(n, m, kind)
    ring.update_tvec(a, lifted)
    ring.update_tvec(a, lifted)                       # twice: dictionaries known the second time
    b = ctx.aggregate(n, m, kind)
    b.update_device(dn, dc)
    b.update_device(dn, dc)
    assert blob_to_dict(a.finalize()) == blob_to_dict(b.finalize())
    want = orc.State(orc.WIDE).update(num, cat, nb=nb).update(num, cat, nb=nb)
    assert blob_to_dict(a.finalize()) == blob_to_dict(want.finalize())
    a.close()
    b.close()


@pytest.mark.parametrize("shape", [((2, 2), (2, 2), False), ((3, 0), (0, 2), False), ((1, 3), (4, 1), False), ((0, 1), (0, 1), False),
                                   ((2, 2), (3, 1), True), ((3, 4), (4, 3), False), ((6, 5), (5, 6), False), ((2, 9), (1, 8), True)])
def test_multiply_kernel_equals_oracle_on_grouped_triples(ctx, shape):
    """multiply_triple over a batch of 500 row pairs of GROUP BY triples (ragged key lists) with
    selection vectors on both sides, device and host entry points."""
    (nA, mA), (nB, mB), nb = shape
    rng = np.random.default_rng(500 + nA + 3 * mB)
    G, rows = 37, 4000
    def side(n, m, seed):
        r = np.random.default_rng(seed)
        gid = r.integers(0, G, rows).astype(np.int32)
        num = [r.integers(0, 8, rows).astype(np.float32) for _ in range(n)]
        cat = [r.integers(-2, 6, rows).astype(np.int32) for _ in range(m)]
        return [st.finalize() for st in orc.grouped_update(num, cat, gid, G, nb=nb)]
    A, B = side(nA, mA, 1), side(nB, mB, 2)
    a_sel, b_sel = rng.integers(0, G, 500), rng.integers(0, G, 500)
    want = [orc.multiply(A[i], B[j], orc.WIDE) for i, j in zip(a_sel, b_sel)]
    for dev in ("cuda", None):
        out = ring.multiply(ctx, ring.tvec_from_blobs(A, device=dev), ring.tvec_from_blobs(B, device=dev), a_sel, b_sel)
        got = out.to_blobs()
        assert len(got) == 500
        for g, w in zip(got, want):
            assert blob_to_dict(g, "num") == blob_to_dict(w, "num")


@pytest.mark.parametrize("n,m,nb,is_key", [(3, 2, False, True), (3, 2, False, False), (0, 2, False, True), (20, 0, False, True),
                                            (4, 3, True, True), (10, 10, False, False)])
def test_group_by_state_pool_equals_per_group_oracle(ctx, n, m, nb, is_key):
    """sum_to_triple ... GROUP BY g with every group in one device table: two batches (new groups
    and new keys in the second, so the table is re-laid out), device and host updates, combine of
    two groups, every group's triple against the oracle's per-row-state update."""
    rng = np.random.default_rng(600 + n + m)
    G, rows = 300, 50_000
    kind = cofactor_hip.NB if nb else cofactor_hip.TRIPLE
    keys = (rng.permutation(100_000)[:G] - 50_000).astype(np.int32) if is_key else np.arange(G, dtype=np.int32)
    slot = np.concatenate([rng.integers(0, G // 2, rows // 2), rng.integers(0, G, rows - rows // 2)]).astype(np.int32)
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    cat = [np.concatenate([rng.integers(0, 3, rows // 2), rng.integers(-40, 40, rows - rows // 2)]).astype(np.int32)
           for _ in range(m)]
    gid = keys[slot]
    grp = ring.Groups(ctx, n, m, kind, is_key=is_key)
    h = rows // 2
    dev = _cuda([gid[:h]]) + _cuda([c[:h] for c in num]) + _cuda([c[:h] for c in cat])
    grp.update_device(dev[0], dev[1:1 + n], dev[1 + n:])
    grp.update_host(gid[h:], [c[h:] for c in num], [c[h:] for c in cat])
    present = np.unique(slot)
    assert grp.count() == (len(present) if is_key else int(slot.max()) + 1)
    want = orc.grouped_update(num, cat, slot, G, nb=nb)
    for s in present[:: max(1, len(present) // 60)]:
        assert blob_to_dict(grp.finalize(int(keys[s]))) == blob_to_dict(want[s].finalize()), s
    a, b = int(present[0]), int(present[-1])
    grp.combine(int(keys[a]), int(keys[b]))
    assert blob_to_dict(grp.finalize(int(keys[a]))) == blob_to_dict(want[a].combine(want[b]).finalize())
    tv, gk = grp.to_tvec("cuda")
    order = np.argsort(keys[present]) if is_key else np.arange(len(present))
    blobs = tv.to_blobs()
    if is_key:
        assert gk.cpu().numpy().tolist() == sorted(keys[present].tolist())
        for r in range(0, len(present), max(1, len(present) // 40)):
            s = present[order[r]]
            assert blob_to_dict(blobs[r]) == blob_to_dict(want[s].finalize()), r
    grp.close()


@pytest.mark.parametrize("n,is_key", [(1, True), (3, False), (7, True), (12, False), (15, True), (16, False), (20, True)])
def test_segmented_group_by_equals_per_group_oracle(n, is_key, monkeypatch):
    """The regroup-then-matrix-core path of wide numeric triples (groupseg.hip), forced on: ragged
    groups (empty ones, one of several work units, one of a single row), a row count that is no
    multiple of anything, two batches (new groups in the second), every group against the oracle."""
    monkeypatch.setenv("COFACTOR_GROUPS_SEG", "1")
    c = cofactor_hip.Context(0)
    try:
        rng = np.random.default_rng(900 + n)
        G, rows = 97, 60_001
        keys = (rng.permutation(10_000)[:G] * 3 - 7_000).astype(np.int32) if is_key else np.arange(G, dtype=np.int32)
        slot = rng.integers(0, G, rows).astype(np.int32)
        slot[slot % 5 == 3] = 11                       # a group of ~12 000 rows: several units; groups = 3 mod 5 stay empty
        slot[:rows // 2][slot[:rows // 2] > 60] = 2    # groups above 60 only appear in the second batch
        slot[-1] = 63                                  # (3 mod 5) a group of exactly one row
        num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
        gid = keys[slot]
        grp = ring.Groups(c, n, 0, cofactor_hip.TRIPLE, is_key=is_key)
        h = rows // 2
        dev = _cuda([gid[:h]]) + _cuda([x[:h] for x in num])
        grp.update_device(dev[0], dev[1:], [])
        grp.update_host(gid[h:], [x[h:] for x in num], [])
        grp.update_device(dev[0], dev[1:], [])         # every group known: the counting pass doubles as the dictionary check
        present = np.unique(slot)
        assert grp.count() == (len(present) if is_key else int(slot.max()) + 1)
        want = orc.grouped_update([np.concatenate([x, x[:h]]) for x in num], [], np.concatenate([slot, slot[:h]]), G, nb=False)
        for s_ in present:
            assert blob_to_dict(grp.finalize(int(keys[s_]))) == blob_to_dict(want[s_].finalize()), s_
        grp.close()
    finally:
        c.close()


@pytest.mark.parametrize("is_key", [True, False])
def test_segmented_group_by_with_more_groups_than_the_lds_histogram_holds(is_key, monkeypatch):
    """40 000 groups: counters and cursors in global memory (the *_atomic kernels of groupseg.hip)."""
    monkeypatch.setenv("COFACTOR_GROUPS_SEG", "1")
    c = cofactor_hip.Context(0)
    try:
        rng = np.random.default_rng(77)
        n, G, rows = 5, 40_000, 300_007
        keys = (rng.permutation(1_000_000)[:G] - 500_000).astype(np.int32) if is_key else np.arange(G, dtype=np.int32)
        slot = rng.integers(0, G, rows).astype(np.int32)
        slot[::3] = 39_999
        num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
        gid = keys[slot]
        grp = ring.Groups(c, n, 0, cofactor_hip.TRIPLE, is_key=is_key)
        dev = _cuda([gid]) + _cuda(num)
        grp.update_device(dev[0], dev[1:], [])
        grp.update_device(dev[0], dev[1:], [])
        want = orc.grouped_update([np.concatenate([x, x]) for x in num], [], np.concatenate([slot, slot]), G, nb=False)
        present = np.unique(slot)
        for s_ in list(present[::400]) + [39_999]:
            assert blob_to_dict(grp.finalize(int(keys[s_]))) == blob_to_dict(want[s_].finalize()), s_
        grp.close()
    finally:
        c.close()


def test_segmented_group_by_matches_the_atomic_path_on_real_values():
    """Non-integer values: the segmented path (fp32 chains of 64 rows folded into fp64, gram.hip's rule)
    against the per-row path (float products summed in fp64) within 1e-6 relative."""
    import os
    import torch
    from cofactor_hip import synth
    n, G, rows = 20, 1000, 1 << 20
    num, _ = synth.table(torch, 5, n, 0, 0, rows, "cuda", keys=4)
    gid = synth.integers(torch, 5, 300, 0, rows, G, "cuda")
    out = []
    for knob in ("1", "2"):
        os.environ["COFACTOR_GROUPS_SEG"] = knob
        try:
            c = cofactor_hip.Context(0)
        finally:
            del os.environ["COFACTOR_GROUPS_SEG"]
        grp = ring.Groups(c, n, 0, cofactor_hip.TRIPLE, is_key=False)
        grp.update_device(gid, num, [])
        out.append([np.asarray(grp.finalize(g)) for g in (0, 1, 500, 999)])
        grp.close()
        c.close()
    for a, b in zip(*out):
        assert a.shape == b.shape and a.size >= 231
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("n,m", [(20, 0), (3, 2)])
def test_group_by_pool_collects_host_chunks_before_it_goes_to_the_device(ctx, n, m):
    """DataChunk-sized host batches (2048 rows) are staged in pinned memory and reach the device 2^18
    rows at a time (and then, for numeric triples, through the segmented path); count / finalize /
    combine in between see every row handed over so far."""
    rng = np.random.default_rng(31 + n)
    G, chunk, nchunks = 40, 2048, 300                     # 614 400 rows: two full staging blocks and a rest
    rows = chunk * nchunks
    slot = rng.integers(0, G, rows).astype(np.int32)
    slot[:chunk] = np.arange(chunk) % 7                   # the first chunk knows seven groups only
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    cat = [rng.integers(-3, 4, rows).astype(np.int32) for _ in range(m)]
    grp = ring.Groups(ctx, n, m, cofactor_hip.TRIPLE, is_key=True)
    keys = (np.arange(G, dtype=np.int32) * 5 - 60)
    gid = keys[slot]
    grp.update_host(gid[:chunk], [c[:chunk] for c in num], [c[:chunk] for c in cat])
    assert grp.count() == 7                               # (flushes what has been staged)
    first = orc.grouped_update([c[:chunk] for c in num], [c[:chunk] for c in cat], slot[:chunk], G, nb=False)
    assert blob_to_dict(grp.finalize(int(keys[3]))) == blob_to_dict(first[3].finalize())
    for i in range(1, nchunks):
        a, b = i * chunk, (i + 1) * chunk
        grp.update_host(gid[a:b], [c[a:b] for c in num], [c[a:b] for c in cat])
    assert grp.count() == G
    want = orc.grouped_update(num, cat, slot, G, nb=False)
    for s_ in range(0, G, 3):
        assert blob_to_dict(grp.finalize(int(keys[s_]))) == blob_to_dict(want[s_].finalize()), s_
    grp.close()


def test_group_by_pool_slot_is_cleared_for_its_next_owner(ctx):
    """cofactor_groups_reset_group: rows staged for the slot's previous owner still go in before the row
    is cleared; the next owner starts from the empty triple; other slots are untouched; key-typed pools
    refuse."""
    rng = np.random.default_rng(4)
    n, m, rows = 3, 1, 5000
    slot = rng.integers(0, 4, rows).astype(np.int32)
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    cat = [rng.integers(0, 5, rows).astype(np.int32) for _ in range(m)]
    grp = ring.Groups(ctx, n, m, cofactor_hip.TRIPLE, is_key=False)
    grp.update_host(slot, num, cat)                  # (staged in pinned memory)
    grp.reset_group(2)
    want = orc.grouped_update(num, cat, slot, 4, nb=False)
    assert blob_to_dict(grp.finalize(1)) == blob_to_dict(want[1].finalize())
    assert blob_to_dict(grp.finalize(2))["N"] == 0
    sel = slot == 0
    grp.update_host(np.full(int(sel.sum()), 2, np.int32), [c[sel] for c in num], [c[sel] for c in cat])
    assert blob_to_dict(grp.finalize(2)) == blob_to_dict(want[0].finalize())
    grp.reset_group(7)                               # (no such row yet: nothing to do)
    grp.close()
    keyed = ring.Groups(ctx, n, m, cofactor_hip.TRIPLE, is_key=True)
    with pytest.raises(cofactor_hip.CofactorError):
        keyed.reset_group(0)
    keyed.close()


def _join_tables(G, per, seed):
    rng = np.random.default_rng(seed)
    rows = G * per
    t = {"gb": np.repeat(np.arange(G, dtype=np.int32) * 7 - 1000, per)}
    perm = rng.permutation(rows)
    t["gb"] = t["gb"][perm]
    for name in "abc":
        t[name] = rng.integers(0, 8, rows).astype(np.float32)
    for name in "def":
        t[name] = rng.integers(0, 4, rows).astype(np.int32)
    return t


def _factorised_join(ctx, t1, t2):
    """select sum_triple(multiply_triple(A, B)) from (select gb, sum_to_triple_2_2(b,c,d,e) A from test1
    group by gb) a join (select gb, sum_to_triple_2_2(a,c,d,f) B from test2 group by gb) b on a.gb = b.gb
    — reference README.md:163-173 — through the C ABI, everything on the device."""
    ga, gb = ring.Groups(ctx, 2, 2, is_key=True), ring.Groups(ctx, 2, 2, is_key=True)
    d1 = _cuda([t1["gb"], t1["b"], t1["c"], t1["d"], t1["e"]])
    d2 = _cuda([t2["gb"], t2["a"], t2["c"], t2["d"], t2["f"]])
    ga.update_device(d1[0], d1[1:3], d1[3:5])
    gb.update_device(d2[0], d2[1:3], d2[3:5])
    A, ka = ga.to_tvec("cuda")
    B, kb = gb.to_tvec("cuda")
    # the join on gb: both key lists are ascending
    _, ia, ib = np.intersect1d(ka.cpu().numpy(), kb.cpu().numpy(), assume_unique=True, return_indices=True)
    prod = ring.multiply(ctx, A, B, ia, ib)
    agg = ctx.aggregate(4, 4)
    ring.update_tvec(agg, prod)
    out = agg.finalize()
    for x in (ga, gb, agg):
        x.close()
    return out, len(ia)


def test_factorised_join_equals_the_aggregate_over_the_joined_table(ctx):
    """2 000 groups x 6 x 5 rows: the factorised result must equal sum_to_triple_4_4 over the
    materialised join (oracle), entry for entry."""
    G = 2000
    t1, t2 = _join_tables(G, 6, 1), _join_tables(G, 5, 2)
    t2["gb"] = t2["gb"] + 7 * 100                     # shift: only part of the groups join
    got, joined_groups = _factorised_join(ctx, t1, t2)
    assert joined_groups == G - 100
    # materialise the join on the host
    o1, o2 = np.argsort(t1["gb"], kind="stable"), np.argsort(t2["gb"], kind="stable")
    k1, k2 = t1["gb"][o1], t2["gb"][o2]
    li, ri = [], []
    for key in np.intersect1d(k1, k2):
        a = o1[np.searchsorted(k1, key, "left"):np.searchsorted(k1, key, "right")]
        b = o2[np.searchsorted(k2, key, "left"):np.searchsorted(k2, key, "right")]
        li.append(np.repeat(a, len(b)))
        ri.append(np.tile(b, len(a)))
    li, ri = np.concatenate(li), np.concatenate(ri)
    num = [t1["b"][li], t1["c"][li], t2["a"][ri], t2["c"][ri]]
    cat = [t1["d"][li], t1["e"][li], t2["d"][ri], t2["f"][ri]]
    want = orc.State(orc.WIDE).update(num, cat).finalize()
    assert blob_to_dict(got) == blob_to_dict(want)


def test_factorised_join_with_100k_groups(ctx):
    """1e5 join keys, 1e6 rows per table: one state pool per table (no per-group buffers), one
    multiply launch over 1e5 row pairs, one sum_triple.  Checked through closed forms of the
    factorised sums computed with numpy from the two tables."""
    G = 100_000
    t1, t2 = _join_tables(G, 10, 3), _join_tables(G, 10, 4)
    got, joined_groups = _factorised_join(ctx, t1, t2)
    assert joined_groups == G
    d = blob_to_dict(got)

    def per_group(t, col):
        return np.bincount((t["gb"] + 1000) // 7, weights=col.astype(np.float64), minlength=G)
    n1, n2 = per_group(t1, np.ones(len(t1["gb"]))), per_group(t2, np.ones(len(t2["gb"])))
    assert d["N"] == float((n1 * n2).sum()) == 1e7
    sb, sc1 = per_group(t1, t1["b"]), per_group(t1, t1["c"])
    sa, sc2 = per_group(t2, t2["a"]), per_group(t2, t2["c"])
    assert d["lin_agg"] == [float((sb * n2).sum()), float((sc1 * n2).sum()), float((sa * n1).sum()), float((sc2 * n1).sum())]
    # quad_agg: [bb, bc, b*a, b*c2, cc, c*a, c*c2, aa, a*c2, c2c2]
    assert d["quad_agg"][2] == float((sb * sa).sum()) and d["quad_agg"][6] == float((sc1 * sc2).sum())
    assert d["quad_agg"][0] == float((per_group(t1, t1["b"] * t1["b"]) * n2).sum())
    # lin_cat of d (table 1): count of key k = sum_g cnt1[g][k] * N2[g]
    for k in range(4):
        want = float((per_group(t1, t1["d"] == k) * n2).sum())
        assert [e["value"] for e in d["lin_cat"][0] if e["key"] == k] == [want]
    # quad_cat (d of table 1) x (f of table 2): pair (k1, k2) = sum_g cnt1[g][k1] cnt2[g][k2]
    pairs = {(e["key1"], e["key2"]): e["value"] for e in d["quad_cat"][2 * 1 + 1]}       # (c1 = 0, c2 = 3) -> index 3
    for k1 in range(4):
        for k2 in range(4):
            want = float((per_group(t1, t1["d"] == k1) * per_group(t2, t2["f"] == k2)).sum())
            assert pairs[(k1, k2)] == want


def test_host_vectors_with_entries_outside_their_extents_are_refused(ctx):
    """The *_host entry points copy the child arrays by their declared extents and the kernels index
    them with the (offset, length) entries as they are: an entry that leaves its extent (a NULL or
    malformed row that came through SQL carries uninitialised list entries) must come back as
    COFACTOR_ERR_INVALID before anything is launched, not as an out-of-bounds device read."""
    rng = np.random.default_rng(5)
    num = [rng.normal(size=50).astype(np.float32) for _ in range(2)]
    cat = [rng.integers(0, 5, 50).astype(np.int32) for _ in range(2)]
    good = ring.lift_host(ctx, num, cat, cofactor_hip.TRIPLE)
    agg = ctx.aggregate(2, 2)
    ring.update_tvec(agg, good)                       # a well-formed vector goes through

    def refused(mutate, restore):
        mutate()
        try:
            with pytest.raises(cofactor_hip.CofactorError) as e:
                ring.update_tvec(agg, good)
            assert e.value.status == cofactor_hip.ERR_INVALID
            with pytest.raises(cofactor_hip.CofactorError) as e:
                ring.multiply(ctx, good, good)
            assert e.value.status == cofactor_hip.ERR_INVALID
        finally:
            restore()

    s = good.struct
    # truncated extents: the last rows' entries now point past them
    for field in ("lin_len", "quad_len", "lc_subs", "nc_subs", "cc_subs", "lc_cap", "nc_cap", "cc_cap"):
        old = getattr(s, field)
        refused(lambda: setattr(s, field, old - 1), lambda: setattr(s, field, old))
    # a wild offset in one sub-list entry / one outer entry / one dense entry
    for name, idx in (("cc_sub", 2 * 7), ("nc_outer", 2 * 3), ("lin_e", 2 * 11), ("lc_sub", 2 * 5 + 1)):
        arr = good.a[name]
        old = int(arr[idx])
        refused(lambda: arr.__setitem__(idx, 1 << 40), lambda: arr.__setitem__(idx, old))
    ring.update_tvec(agg, good)                       # untouched again: accepted
    want = orc.State(orc.WIDE).update(num, cat).update(num, cat)
    assert blob_to_dict(agg.finalize()) == blob_to_dict(want.finalize())
    agg.close()
