"""The triple's consumers: linreg_train / lda_train of libcofactor_hip (host
fp64) against the numpy restatement in oracle/ml_oracle.py and against sklearn, in the
reference's own scenarios (duckdb_extension/test/python/test_regression.py:96-163,
test_LDA.py:97-197: iris, test_size=0.33, random_state=42, "same score as sklearn to 3
decimals").  Training never touches the GPU, so these run on CPU; the predict kernels are
checked in the gpu tests at the bottom."""
import numpy as np
import pytest

import cofactor_hip
from oracle import ml_oracle, oracle
from triple_fmt import blob_to_dict

COLS = ["s_length", "s_width", "p_length", "p_width"]


def _iris(binned=False):
    from sklearn.datasets import load_iris
    from sklearn.model_selection import train_test_split
    from sklearn.preprocessing import KBinsDiscretizer
    X, y = load_iris(as_frame=True, return_X_y=True)
    X.columns = COLS
    if binned:       # test_LDA.py:57-62 / test_regression.py:58-63
        est = KBinsDiscretizer(n_bins=4, encode="ordinal", strategy="uniform", subsample=None)
        b = est.fit_transform(X[["s_length", "s_width", "p_length"]])
        for j, c in enumerate(["s_length", "s_width", "p_length"]):
            X[c] = b[:, j]
    tr, te, ytr, yte = train_test_split(X, y, test_size=0.33, random_state=42)
    tr = tr.assign(target=ytr)
    te = te.assign(target=yte)
    return tr, te


def _cols(df, num, cat):
    return ([df[c].to_numpy(dtype=np.float32) for c in num],
            [df[c].to_numpy().astype(np.int32) for c in cat])


def _triple(df, num, cat):
    n, c = _cols(df, num, cat)
    return oracle.State(oracle.WIDE).update(n, c).finalize()


def _close(a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.allclose(a, b, rtol=tol, atol=tol), np.abs(a - b).max()


# ---- linear regression ------------------------------------------------------------------------

@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("variance", [False, True])
def test_linreg_train_matches_restatement(normalize, variance):
    tr, _ = _iris()
    blob = _triple(tr, COLS, ["target"])
    got = cofactor_hip.linreg_train(blob, 0, 0.001, 0.0, 10000, variance, normalize)
    want = ml_oracle.linreg_train(blob_to_dict(blob), 0, 0.001, 0.0, 10000, variance, normalize)
    _close(got, want, 2e-4)
    assert got[0] == 1 and list(got[1:6]) == [0, 3, 0, 1, 2]      # m, begin[0..1], keys


def test_linreg_train_ridge_and_binned_keys_match_restatement():
    tr, _ = _iris(binned=True)
    blob = _triple(tr, ["p_width", "p_length"], ["s_length", "s_width", "target"])
    for lam in (0.0, 0.1):
        got = cofactor_hip.linreg_train(blob, 1, 0.001, lam, 10000, True, False)
        want = ml_oracle.linreg_train(blob_to_dict(blob), 1, 0.001, lam, 10000, True, False)
        _close(got, want, 2e-4)


@pytest.mark.parametrize("normalize", [False, True])
def test_linreg_matches_sklearn_r2(normalize):
    """test_regression.py::test_linreg_no_norm / test_linreg_norm"""
    import pandas as pd
    from sklearn.linear_model import LinearRegression
    from sklearn.metrics import r2_score
    tr, te = _iris()
    blob = _triple(tr, COLS, ["target"])
    params = cofactor_hip.linreg_train(blob, 0, 0.001, 0.0, 10000, False, normalize)
    n, c = _cols(te, COLS[1:], ["target"])
    pred = ml_oracle.linreg_predict(params, False, normalize, n, c)
    r2 = r2_score(te["s_length"], pred)
    tre, tee = pd.get_dummies(tr, columns=["target"]), pd.get_dummies(te, columns=["target"])
    reg = LinearRegression().fit(tre.drop(columns=["s_length"]), tre["s_length"])
    assert round(r2, 3) == round(reg.score(tee.drop(columns=["s_length"]), tee["s_length"]), 3)


def test_linreg_train_rejects_bad_input():
    tr, _ = _iris()
    blob = _triple(tr, COLS, ["target"])
    with pytest.raises(cofactor_hip.CofactorError):
        cofactor_hip.linreg_train(blob, 4)                 # label out of range
    nb = oracle.State(oracle.WIDE).update(*_cols(tr, COLS, ["target"]), nb=True).finalize()
    with pytest.raises(cofactor_hip.CofactorError):
        cofactor_hip.linreg_train(nb, 0)                   # nb aggregate has no cross terms


# ---- LDA ----------------------------------------------------------------------------------------

@pytest.mark.parametrize("normalize", [False, True])
def test_lda_train_matches_restatement_and_sklearn(normalize):
    """test_LDA.py::test_lda_no_norm / test_lda_norm"""
    from sklearn.discriminant_analysis import LinearDiscriminantAnalysis
    tr, te = _iris()
    blob = _triple(tr, COLS, ["target"])
    got = cofactor_hip.lda_train(blob, 0, 0.0, normalize)
    want = ml_oracle.lda_train(blob_to_dict(blob), 0, 0.0, normalize)
    _close(got, want, 1e-4)
    assert list(got[:2]) == [3, 0] and list(got[2:5]) == [0, 1, 2]
    n, c = _cols(te, COLS, [])
    pred = ml_oracle.lda_predict(got, normalize, n, c)
    clf = LinearDiscriminantAnalysis(solver="lsqr", shrinkage=0).fit(tr[COLS], tr["target"])
    assert round(float((pred == te["target"]).mean()), 3) == round(clf.score(te[COLS], te["target"]), 3)


@pytest.mark.parametrize("normalize", [False, True])
def test_lda_with_key_features_matches_restatement_and_sklearn(normalize):
    """test_LDA.py::test_lda_no_norm_cat / test_lda_norm_cat: sum_to_triple_1_4, label 3"""
    import pandas as pd
    from sklearn.discriminant_analysis import LinearDiscriminantAnalysis
    tr, te = _iris(binned=True)
    cats = ["s_length", "s_width", "p_length", "target"]
    blob = _triple(tr, ["p_width"], cats)
    got = cofactor_hip.lda_train(blob, 3, 0.01, normalize)
    want = ml_oracle.lda_train(blob_to_dict(blob), 3, 0.01, normalize)
    _close(got, want, 1e-3)
    n, c = _cols(te, ["p_width"], cats[:3])
    pred = ml_oracle.lda_predict(got, normalize, n, c)
    full = pd.get_dummies(pd.concat([tr, te]), columns=cats[:3])
    tre, tee = full.iloc[:len(tr)], full.iloc[len(tr):]
    clf = LinearDiscriminantAnalysis(solver="lsqr", shrinkage=0)
    clf.fit(tre.drop(columns=["target"]), tre["target"])
    acc = clf.score(tee.drop(columns=["target"]), tee["target"])
    assert round(float((pred == te["target"]).mean()), 3) == round(acc, 3)


def test_lda_label_in_the_middle_uses_one_layout():
    """label not the last key column: coefficients line up with the predictor's one-hot layout"""
    tr, te = _iris(binned=True)
    cats = ["s_length", "target", "s_width"]
    blob = _triple(tr, ["p_width", "p_length"], cats)
    got = cofactor_hip.lda_train(blob, 1, 0.01, False)
    want = ml_oracle.lda_train(blob_to_dict(blob), 1, 0.01, False)
    _close(got, want, 1e-3)
    n, c = _cols(te, ["p_width", "p_length"], ["s_length", "s_width"])
    pred = ml_oracle.lda_predict(got, False, n, c)
    assert (pred == te["target"]).mean() > 0.9


def test_min_norm_solve_of_a_singular_system():
    """shrinkage 0 with key features: the pooled covariance is singular (each one-hot block sums
    to one); lda_train must return the minimum-norm solution like dgelsd, not blow up"""
    tr, _ = _iris(binned=True)
    blob = _triple(tr, ["p_width"], ["s_length", "s_width", "target"])
    got = cofactor_hip.lda_train(blob, 2, 0.0, False)
    want = ml_oracle.lda_train(blob_to_dict(blob), 2, 0.0, False)
    assert np.all(np.isfinite(got))
    _close(got, want, 5e-3)


# ---- predict kernels (GPU) ----------------------------------------------------------------------

def _dev(cols):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(c)).cuda() for c in cols]


@pytest.mark.gpu
@pytest.mark.parametrize("normalize", [False, True])
def test_linreg_predict_kernel_matches_restatement(normalize):
    import torch
    tr, te = _iris(binned=True)
    num, cat = ["p_width", "p_length"], ["s_length", "s_width", "target"]
    ctx = cofactor_hip.Context(0)
    agg = ctx.aggregate(2, 3)
    n, c = _cols(tr, num, cat)
    agg.update_device(_dev(n), _dev(c))
    params = cofactor_hip.linreg_train(agg.finalize(), 1, 0.001, 0.0, 10000, True, normalize)
    tn, tc = _cols(te, num[:1], cat)
    want = ml_oracle.linreg_predict(params, False, normalize, tn, tc)
    out = torch.zeros(len(te), dtype=torch.float32, device="cuda")
    ctx.linreg_predict(params, _dev(tn), _dev(tc), out=out, normalize=normalize)
    _close(out.cpu().numpy(), want, 1e-6)
    _close(ctx.linreg_predict(params, tn, tc, normalize=normalize), want, 1e-6)   # host columns
    # masked in-place update: CASE WHEN is_null THEN predict ELSE col END
    col = torch.from_numpy(te["p_length"].to_numpy(dtype=np.float32)).cuda()
    mask = torch.zeros(len(te), dtype=torch.uint8, device="cuda")
    mask[::3] = 1
    before = col.clone()
    ctx.linreg_predict(params, _dev(tn), _dev(tc), out=col, mask=mask, normalize=normalize)
    got = col.cpu().numpy()
    keep = mask.cpu().numpy() == 0
    assert np.array_equal(got[keep], before.cpu().numpy()[keep])
    _close(got[~keep], want[~keep], 1e-6)
    # a key the model never saw adds nothing
    tc2 = [a.copy() for a in tc]
    tc2[0][:] = 77
    w2 = ml_oracle.linreg_predict(params, False, normalize, tn, tc2)
    _close(ctx.linreg_predict(params, tn, tc2, normalize=normalize), w2, 1e-6)
    agg.close(); ctx.close()


@pytest.mark.gpu
def test_linreg_predict_noise_is_gaussian_and_reproducible():
    import torch
    tr, _ = _iris()
    blob = _triple(tr, COLS, ["target"])
    params = cofactor_hip.linreg_train(blob, 0, 0.001, 0.0, 10000, True, False)
    rows = 1 << 20
    rng = np.random.default_rng(5)
    n = [rng.random(rows, dtype=np.float32) for _ in range(3)]
    c = [rng.integers(0, 3, rows).astype(np.int32)]
    ctx = cofactor_hip.Context(0)
    dn, dc = _dev(n), _dev(c)
    base = torch.empty(rows, dtype=torch.float32, device="cuda")
    a = torch.empty_like(base); b = torch.empty_like(base); d = torch.empty_like(base)
    ctx.linreg_predict(params, dn, dc, out=base)
    ctx.linreg_predict(params, dn, dc, out=a, noise=True, seed=1)
    ctx.linreg_predict(params, dn, dc, out=b, noise=True, seed=1)
    ctx.linreg_predict(params, dn, dc, out=d, noise=True, seed=2)
    assert torch.equal(a, b) and not torch.equal(a, d)
    z = ((a - base) / float(params[-1])).double().cpu().numpy()
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3
    assert abs(np.mean(z ** 3)) < 2e-2 and abs(np.mean(z ** 4) - 3) < 5e-2
    # shard-independent: predicting the second half alone gives the same values only if the
    # generator keys on the global row, which a shard does not know — it keys on (seed, row)
    # of the call, so shards take distinct seeds; the two halves must not repeat each other
    assert not torch.equal(a[: rows // 2], a[rows // 2:])
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("normalize", [False, True])
def test_lda_predict_kernel_matches_restatement(normalize):
    import torch
    tr, te = _iris(binned=True)
    cats = ["s_length", "target", "s_width"]
    ctx = cofactor_hip.Context(0)
    agg = ctx.aggregate(2, 3)
    n, c = _cols(tr, ["p_width", "p_length"], cats)
    agg.update_device(_dev(n), _dev(c))
    params = cofactor_hip.lda_train(agg.finalize(), 1, 0.01, normalize)
    tn, tc = _cols(te, ["p_width", "p_length"], ["s_length", "s_width"])
    want = ml_oracle.lda_predict(params, normalize, tn, tc)
    out = torch.full((len(te),), -1, dtype=torch.int32, device="cuda")
    ctx.lda_predict(params, _dev(tn), _dev(tc), out=out, normalize=normalize)
    assert np.array_equal(out.cpu().numpy(), want)
    assert np.array_equal(ctx.lda_predict(params, tn, tc, normalize=normalize), want)
    labels = np.array(ml_oracle.lda_labels(params), dtype=np.int32)
    assert np.array_equal(ctx.lda_predict(params, tn, tc, normalize=normalize, emit_label=True),
                          labels[want])
    assert (want == te["target"]).mean() > 0.9
    agg.close(); ctx.close()


@pytest.mark.gpu
def test_predict_rejects_a_parameter_vector_of_another_shape():
    tr, te = _iris()
    blob = _triple(tr, COLS, ["target"])
    params = cofactor_hip.linreg_train(blob, 0)
    tn, tc = _cols(te, COLS[1:], ["target"])
    ctx = cofactor_hip.Context(0)
    with pytest.raises(cofactor_hip.CofactorError):
        ctx.linreg_predict(params, tn[:1], tc)
    with pytest.raises(cofactor_hip.CofactorError):
        ctx.linreg_predict(params, tn, [])
    with pytest.raises(cofactor_hip.CofactorError):
        ctx.linreg_predict(params, tn, tc, noise=True)     # trained without variance
    ctx.close()


# ---- edge cases of the trainers (CPU) ---------------------------------------------------------------

def test_trainers_on_degenerate_triples():
    """no key columns at all, one class only, a label that is not a column: clean errors or finite
    parameter vectors, never a crash (the reference asserts or reads out of bounds here)."""
    rng = np.random.default_rng(1)
    x = [rng.normal(size=50).astype(np.float32) for _ in range(3)]
    only_num = oracle.State(oracle.WIDE).update(x, []).finalize()
    p = cofactor_hip.linreg_train(only_num, 2, 0.001, 0.0, 500, True, False)
    assert p[0] == 0 and len(p) == 1 + 3 + 1 and np.all(np.isfinite(p))      # m, intercept + 2 coefs, std
    with pytest.raises(cofactor_hip.CofactorError):
        cofactor_hip.lda_train(only_num, 0)                                   # no key column to classify
    one_class = oracle.State(oracle.WIDE).update(x, [np.full(50, 4, dtype=np.int32)]).finalize()
    q = cofactor_hip.lda_train(one_class, 0, 0.1, False)
    assert q[0] == 1 and q[2] == 4 and np.all(np.isfinite(q))                 # one class, its key
    with pytest.raises(cofactor_hip.CofactorError):
        cofactor_hip.lda_train(one_class, 1)
    with pytest.raises(cofactor_hip.CofactorError):
        cofactor_hip.linreg_train(np.zeros(3), 0)                             # not a triple blob


def test_linreg_gradient_descent_reaches_the_normal_equations():
    """the reference's descent, run long enough on a well-conditioned problem, solves X^T X w = X^T y"""
    rng = np.random.default_rng(2)
    rows = 400
    x1, x2 = rng.normal(size=rows), rng.normal(size=rows)
    y = 1.5 + 2.0 * x1 - 0.5 * x2 + 0.05 * rng.normal(size=rows)
    cols = [c.astype(np.float32) for c in (y, x1, x2)]
    blob = oracle.State(oracle.WIDE).update(cols, []).finalize()
    p = cofactor_hip.linreg_train(blob, 0, 0.001, 0.0, 10000, False, False)
    X = np.stack([np.ones(rows), cols[1].astype(np.float64), cols[2].astype(np.float64)], 1)
    w = np.linalg.lstsq(X, cols[0].astype(np.float64), rcond=None)[0]
    assert np.allclose(p[1:], w, rtol=1e-3, atol=1e-3)


def test_two_call_protocol_trains_once_and_never_mixes_up_calls():
    """The C ABI's size query has to train; the call with the buffer that follows returns that very
    result (per-thread memo keyed by the triple and every argument).  Another label, another triple
    or another flag must not hit it."""
    import ctypes as C
    import time
    tr, _ = _iris()
    blob = np.ascontiguousarray(_triple(tr, COLS, ["target"]), dtype=np.float64)
    L = cofactor_hip.lib()

    def raw(b, label, variance, cap):
        need = C.c_uint64(0)
        out = np.zeros(max(cap, 1), dtype=np.float32)
        st = L.cofactor_linreg_train(b.ctypes.data, b.size, label, 0.001, 0.0, 10000, int(variance), 0,
                                     out.ctypes.data if cap else None, cap, C.byref(need))
        return st, need.value, out[:need.value] if cap >= need.value else None

    st, need, _ = raw(blob, 0, False, 0)
    assert st == cofactor_hip.OK and need > 0
    t0 = time.perf_counter()
    st, need2, first = raw(blob, 0, False, need)
    dt = time.perf_counter() - t0
    assert st == cofactor_hip.OK and need2 == need
    want = ml_oracle.linreg_train(blob_to_dict(blob), 0, 0.001, 0.0, 10000, False, False)
    _close(first, want, 2e-4)
    assert dt < 0.01                                    # no second training
    other = cofactor_hip.linreg_train(blob, 1, 0.001, 0.0, 10000, False, False)
    assert not np.allclose(other, first)
    _close(other, ml_oracle.linreg_train(blob_to_dict(blob), 1, 0.001, 0.0, 10000, False, False), 2e-4)
    with_var = cofactor_hip.linreg_train(blob, 0, 0.001, 0.0, 10000, True, False)
    assert with_var.size == first.size + 1
    changed = blob.copy()
    changed[3] += 1.0                                   # (N)
    assert not np.allclose(cofactor_hip.linreg_train(changed, 0, 0.001, 0.0, 10000, False, False), first)
    assert raw(blob, 0, False, 1)[0] == cofactor_hip.ERR_CAPACITY
