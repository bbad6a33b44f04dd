"""Edge shapes of round 3's kernels against the oracle: segmented GROUP BY (groupseg.hip) with one
group, one row per group, fewer rows than a wave, every n; multiply (mulfill.hip) with one row, no
selection vectors, one-sided shapes and every lanes-per-row variant."""
import numpy as np
import pytest

import cofactor_hip
from cofactor_hip import ring
from oracle import oracle as orc
from triple_fmt import blob_to_dict

pytestmark = pytest.mark.gpu


def _cuda(cols):
    import torch
    out = [torch.from_numpy(np.ascontiguousarray(c)).cuda() for c in cols]
    torch.cuda.synchronize()
    return out


@pytest.fixture(scope="module")
def seg_ctx():
    import os
    os.environ["COFACTOR_GROUPS_SEG"] = "1"
    try:
        c = cofactor_hip.Context(0)
    finally:
        del os.environ["COFACTOR_GROUPS_SEG"]
    yield c
    c.close()


@pytest.mark.parametrize("n", list(range(1, 21)))
def test_segmented_group_by_every_width(seg_ctx, n):
    rng = np.random.default_rng(n)
    G, rows = 7, 5_003
    slot = rng.integers(0, G, rows).astype(np.int32)
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    grp = ring.Groups(seg_ctx, n, 0, cofactor_hip.TRIPLE, is_key=False)
    dev = _cuda([slot]) + _cuda(num)
    grp.update_device(dev[0], dev[1:], [])
    want = orc.grouped_update(num, [], slot, G, nb=False)
    for s_ in range(G):
        assert blob_to_dict(grp.finalize(s_)) == blob_to_dict(want[s_].finalize()), s_
    grp.close()


@pytest.mark.parametrize("case", ["one_group", "one_row_each", "few_rows", "one_row", "all_but_one"])
@pytest.mark.parametrize("is_key", [True, False])
def test_segmented_group_by_degenerate_batches(seg_ctx, case, is_key):
    rng = np.random.default_rng(len(case))
    n = 6
    if case == "one_group":
        G, rows = 1, 9_000
        slot = np.zeros(rows, np.int32)
    elif case == "one_row_each":
        G, rows = 3_000, 3_000
        slot = rng.permutation(G).astype(np.int32)
    elif case == "few_rows":
        G, rows = 5, 37
        slot = rng.integers(0, G, rows).astype(np.int32)
    elif case == "one_row":
        G, rows = 1, 1
        slot = np.zeros(1, np.int32)
    else:
        G, rows = 50, 20_000
        slot = np.full(rows, 17, np.int32)
        slot[::997] = rng.integers(0, G, len(slot[::997]))
    keys = (np.arange(G, dtype=np.int32) * 11 - 300) if is_key else np.arange(G, dtype=np.int32)
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    grp = ring.Groups(seg_ctx, n, 0, cofactor_hip.TRIPLE, is_key=is_key)
    dev = _cuda([keys[slot]]) + _cuda(num)
    grp.update_device(dev[0], dev[1:], [])
    grp.update_device(dev[0], dev[1:], [])
    want = orc.grouped_update([np.concatenate([x, x]) for x in num], [], np.concatenate([slot, slot]), G, nb=False)
    present = np.unique(slot)
    for s_ in present[:: max(1, len(present) // 50)]:
        assert blob_to_dict(grp.finalize(int(keys[s_]))) == blob_to_dict(want[s_].finalize()), s_
    grp.close()


@pytest.mark.parametrize("gl", ["8", "16", "32", "64"])
@pytest.mark.parametrize("rows", [1, 5, 64, 333])
def test_multiply_every_lane_group_width(gl, rows, monkeypatch):
    """COFACTOR_MUL_GL pins the lanes per output row of mul_fill_kernel (read once per process: the
    first call fixes it, so every variant runs in its own interpreter)."""
    import subprocess
    import sys
    import os
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import cofactor_hip
from cofactor_hip import ring
from oracle import oracle as orc
from triple_fmt import blob_to_dict
rows = %d
ctx = cofactor_hip.Context(0)
G = 9
def side(n, m, seed):
    r = np.random.default_rng(seed)
    gid = r.integers(0, G, 600).astype(np.int32)
    num = [r.integers(0, 8, 600).astype(np.float32) for _ in range(n)]
    cat = [r.integers(-2, 5, 600).astype(np.int32) for _ in range(m)]
    return [st.finalize() for st in orc.grouped_update(num, cat, gid, G, nb=False)]
A, B = side(2, 3, 1), side(3, 2, 2)
rng = np.random.default_rng(rows)
a_sel, b_sel = rng.integers(0, G, rows), rng.integers(0, G, rows)
want = [orc.multiply(A[i], B[j], orc.WIDE) for i, j in zip(a_sel, b_sel)]
out = ring.multiply(ctx, ring.tvec_from_blobs(A, device="cuda"), ring.tvec_from_blobs(B, device="cuda"), a_sel, b_sel)
got = out.to_blobs()
assert len(got) == rows
for g, w in zip(got, want):
    assert blob_to_dict(g) == blob_to_dict(w)
if rows == G:
    pass
ctx.close()
print("OK")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, COFACTOR_MUL_GL=gl)
    out = subprocess.run([sys.executable, "-c", code % (os.path.join(root, "duckdb-imputation_amd"), root,
                                                        os.path.join(root, "tests"), rows)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_multiply_without_selection_vectors_and_with_one_sided_shapes():
    ctx = cofactor_hip.Context(0)
    G = 40
    def side(n, m, seed):
        r = np.random.default_rng(seed)
        gid = r.integers(0, G, 3000).astype(np.int32)
        gid[:G] = np.arange(G)
        num = [r.integers(0, 8, 3000).astype(np.float32) for _ in range(n)]
        cat = [r.integers(-2, 5, 3000).astype(np.int32) for _ in range(m)]
        return [st.finalize() for st in orc.grouped_update(num, cat, gid, G, nb=False)]
    for (sa, sb) in (((3, 0), (0, 0)), ((0, 0), (0, 3)), ((0, 2), (2, 0)), ((1, 1), (0, 0))):
        A, B = side(*sa, 3), side(*sb, 4)
        out = ring.multiply(ctx, ring.tvec_from_blobs(A, device="cuda"), ring.tvec_from_blobs(B, device="cuda"))
        got = out.to_blobs()
        assert len(got) == G
        for g, a, b in zip(got, A, B):
            assert blob_to_dict(g) == blob_to_dict(orc.multiply(a, b, orc.WIDE)), (sa, sb)
    ctx.close()
