"""Flat triple blob <-> the nested dict DuckDB's Python client returns for a triple STRUCT.

Test-side helper (independent of the product's own decoder).  Blob layout: see
include/cofactor_hip.h ("flat triple blob") — it follows the order in which the reference's
finalize fills its nested vectors (duckdb_extension/src/triple/sum/sum_state.cpp:116-464).
"""
import numpy as np


def tri(k):
    return k * (k + 1) // 2


def blob_to_dict(blob, names="agg"):
    """names='agg' -> lin_agg/quad_agg (aggregates, sum_no_lift.cpp:23-24);
    names='num' -> lin_num/quad_num (scalars, lift.cpp:256-257)."""
    b = np.asarray(blob, dtype=np.float64)
    kind, n, m = int(b[0]), int(b[1]), int(b[2])
    out = {"N": int(b[3])}
    p = 4
    out["lin_" + names] = [float(x) for x in b[p:p + n]]
    p += n
    qn = n if kind else tri(n)
    out["quad_" + names] = [float(x) for x in b[p:p + qn]]
    p += qn

    def kv_lists(count):
        nonlocal p
        res = []
        for _ in range(count):
            ln = int(b[p]); p += 1
            res.append([{"key": int(b[p + 2 * e]), "value": float(b[p + 2 * e + 1])}
                        for e in range(ln)])
            p += 2 * ln
        return res

    out["lin_cat"] = kv_lists(m)
    if kind == 0:
        out["quad_num_cat"] = kv_lists(n * m)
        qc = []
        for _ in range(tri(m)):
            ln = int(b[p]); p += 1
            qc.append([{"key1": int(b[p + 3 * e]), "key2": int(b[p + 3 * e + 1]),
                        "value": float(b[p + 3 * e + 2])} for e in range(ln)])
            p += 3 * ln
        out["quad_cat"] = qc
    assert p == len(b), (p, len(b))
    return out


def blob_sections(blob):
    """The same walk without a Python object per entry (blobs with 1e7+ entries): returns
    (header dict, lin_cat, quad_num_cat, quad_cat) where every list is a numpy view of shape
    [entries, 2] (key, value) or [entries, 3] (key1, key2, value)."""
    b = np.asarray(blob, dtype=np.float64)
    kind, n, m = int(b[0]), int(b[1]), int(b[2])
    p = 4
    head = {"kind": kind, "n": n, "m": m, "N": int(b[3]), "lin": b[p:p + n]}
    p += n
    qn = n if kind else tri(n)
    head["quad"] = b[p:p + qn]
    p += qn

    def lists(count, width):
        nonlocal p
        res = []
        for _ in range(count):
            ln = int(b[p]); p += 1
            res.append(b[p:p + width * ln].reshape(ln, width))
            p += width * ln
        return res

    lin_cat = lists(m, 2)
    num_cat = lists(n * m, 2) if kind == 0 else []
    cat_cat = lists(tri(m), 3) if kind == 0 else []
    assert p == len(b), (p, len(b))
    return head, lin_cat, num_cat, cat_cat


def dict_to_blob(d, kind=None):
    names = "agg" if "lin_agg" in d else "num"
    if kind is None:
        kind = 0 if "quad_cat" in d else 1
    lin, quad = d["lin_" + names], d["quad_" + names]
    n, m = len(lin), len(d["lin_cat"])
    out = [kind, n, m, d["N"]] + list(lin) + list(quad)
    for lst in d["lin_cat"]:
        out.append(len(lst))
        for e in lst:
            out += [e["key"], e["value"]]
    if kind == 0:
        for lst in d["quad_num_cat"]:
            out.append(len(lst))
            for e in lst:
                out += [e["key"], e["value"]]
        for lst in d["quad_cat"]:
            out.append(len(lst))
            for e in lst:
                out += [e["key1"], e["key2"], e["value"]]
    return np.array(out, dtype=np.float64)


def dense_truth(num_cols, cat_cols, nb=False):
    """Independent numpy/fp64 statement of the triple (no loops over rows): used to
    cross-check the oracle itself on random data."""
    X = np.stack([np.asarray(c, dtype=np.float64) for c in num_cols], axis=1) if num_cols \
        else np.zeros((len(cat_cols[0]) if cat_cols else 0, 0))
    n, m = X.shape[1], len(cat_cols)
    rows = X.shape[0] if n else (len(cat_cols[0]) if m else 0)
    G = X.T @ X
    quad = [G[j, j] for j in range(n)] if nb else [G[j, k] for j in range(n) for k in range(j, n)]
    d = {"N": rows, "lin_agg": list(X.sum(axis=0)), "quad_agg": quad, "lin_cat": []}
    keys = []
    for c in cat_cols:
        c = np.asarray(c)
        u, inv, cnt = np.unique(c, return_inverse=True, return_counts=True)
        keys.append((u, inv))
        d["lin_cat"].append([{"key": int(k), "value": float(v)} for k, v in zip(u, cnt)])
    if not nb:
        d["quad_num_cat"] = []
        for k in range(n):
            for (u, inv) in keys:
                s = np.bincount(inv, weights=X[:, k], minlength=len(u))
                d["quad_num_cat"].append([{"key": int(a), "value": float(b)} for a, b in zip(u, s)])
        d["quad_cat"] = []
        for c1 in range(m):
            for c2 in range(c1, m):
                pair = np.stack([np.asarray(cat_cols[c1]), np.asarray(cat_cols[c2])], axis=1)
                u, cnt = np.unique(pair, axis=0, return_counts=True)
                d["quad_cat"].append([{"key1": int(a), "key2": int(b), "value": float(v)}
                                      for (a, b), v in zip(u, cnt)])
    return d


def assert_triple_close(got, want, rtol=0.0, atol=0.0, exact_counts=True):
    """Compare two nested dicts: keys, N and counts exactly; float sums within rtol/atol."""
    assert got["N"] == want["N"]
    names = "agg" if "lin_agg" in want else "num"
    gn = "agg" if "lin_agg" in got else "num"

    def close(a, b, what):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        assert a.shape == b.shape, (what, a.shape, b.shape)
        if a.size:
            err = np.abs(a - b)
            tol = atol + rtol * np.abs(b)
            assert np.all(err <= tol), (what, float(err.max()), a[np.argmax(err - tol)],
                                        b[np.argmax(err - tol)])

    close(got["lin_" + gn], want["lin_" + names], "lin")
    close(got["quad_" + gn], want["quad_" + names], "quad")
    assert len(got["lin_cat"]) == len(want["lin_cat"])
    for g, w in zip(got["lin_cat"], want["lin_cat"]):
        assert [e["key"] for e in g] == [e["key"] for e in w]
        if exact_counts:
            assert [e["value"] for e in g] == [e["value"] for e in w]
        else:
            close([e["value"] for e in g], [e["value"] for e in w], "lin_cat")
    if "quad_num_cat" in want:
        assert len(got["quad_num_cat"]) == len(want["quad_num_cat"])
        for g, w in zip(got["quad_num_cat"], want["quad_num_cat"]):
            assert [e["key"] for e in g] == [e["key"] for e in w]
            close([e["value"] for e in g], [e["value"] for e in w], "quad_num_cat")
        assert len(got["quad_cat"]) == len(want["quad_cat"])
        for g, w in zip(got["quad_cat"], want["quad_cat"]):
            assert [(e["key1"], e["key2"]) for e in g] == [(e["key1"], e["key2"]) for e in w]
            if exact_counts:
                assert [e["value"] for e in g] == [e["value"] for e in w]
            else:
                close([e["value"] for e in g], [e["value"] for e in w], "quad_cat")
