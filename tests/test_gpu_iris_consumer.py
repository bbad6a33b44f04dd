"""The triple as its consumers use it (SURVEY.md §4 "Statistical end-to-end, ML"): the reference's
test_regression.py::test_linreg_no_norm scenario.  Iris, the reference's split (test_size=0.33,
random_state=42), the triple of sum_to_triple_4_1(s_length, s_width, p_length, p_width, target)
computed by the HIP path; ordinary least squares solved from the triple alone (the sigma matrix of
ML/utils.cpp:176-310: intercept, numeric columns, one-hot key columns) must predict the test set
with the same R^2 (3 decimals) as sklearn's LinearRegression on the one-hot encoded rows — the
assertion the reference makes for linreg_train/linreg_predict (test_regression.py:141)."""
import numpy as np
import pytest

import cofactor_hip
from triple_fmt import blob_to_dict

pytestmark = pytest.mark.gpu


def sigma_from_triple(t, n, m):
    """(X^T X over [1, x_0..x_{n-1}, onehot(c_0), ..]) and the column layout, from the triple only."""
    keys = [[e["key"] for e in lst] for lst in t["lin_cat"]]
    off = [1 + n]
    for k in keys:
        off.append(off[-1] + len(k))
    p = off[-1]
    S = np.zeros((p, p))
    S[0, 0] = t["N"]
    S[0, 1:1 + n] = S[1:1 + n, 0] = t["lin_agg"]
    q = 0
    for j in range(n):
        for k in range(j, n):
            S[1 + j, 1 + k] = S[1 + k, 1 + j] = t["quad_agg"][q]
            q += 1
    for c in range(m):
        for i, e in enumerate(t["lin_cat"][c]):
            S[0, off[c] + i] = S[off[c] + i, 0] = e["value"]
        for j in range(n):
            for i, e in enumerate(t["quad_num_cat"][j * m + c]):
                S[1 + j, off[c] + i] = S[off[c] + i, 1 + j] = e["value"]
    q = 0
    for c1 in range(m):
        for c2 in range(c1, m):
            for e in t["quad_cat"][q]:
                i1 = off[c1] + keys[c1].index(e["key1"])
                i2 = off[c2] + keys[c2].index(e["key2"])
                S[i1, i2] = S[i2, i1] = e["value"]
            q += 1
    return S, keys, off


def test_iris_linear_regression_from_gpu_triple_matches_sklearn():
    import pandas as pd
    import torch
    from sklearn.datasets import load_iris
    from sklearn.linear_model import LinearRegression
    from sklearn.metrics import r2_score
    from sklearn.model_selection import train_test_split

    data = load_iris(as_frame=True, return_X_y=True)
    df_train, df_test, y_train, y_test = train_test_split(data[0], data[1], test_size=0.33, random_state=42)
    cols = ["sepal length (cm)", "sepal width (cm)", "petal length (cm)", "petal width (cm)"]
    num = [df_train[c].to_numpy(dtype=np.float32) for c in cols]      # FLOAT columns, as in DuckDB
    cat = [y_train.to_numpy(dtype=np.int32)]
    ctx = cofactor_hip.Context(0)
    agg = ctx.aggregate(4, 1)
    dn = [torch.from_numpy(c).cuda() for c in num]
    dc = [torch.from_numpy(c).cuda() for c in cat]
    torch.cuda.synchronize()
    agg.update_device(dn, dc)
    t = blob_to_dict(agg.finalize())
    agg.close(); ctx.close()
    assert t["N"] == len(df_train)

    S, keys, off = sigma_from_triple(t, 4, 1)
    label = 1                                              # s_length: column 1 of [1, x0..x3, onehot]
    feat = [i for i in range(S.shape[0]) if i != label]
    # intercept + a full one-hot block is rank deficient by construction (smallest true singular
    # value ratio here is 3e-4, the null one is rounding noise ~1e-8): cut at 1e-6, the min-norm
    # solution predicts like the reference's gradient descent (ML/regression.cpp:176-238) does
    w = np.linalg.lstsq(S[np.ix_(feat, feat)], S[feat, label], rcond=1e-6)[0]
    Xt = np.zeros((len(df_test), S.shape[0]))
    Xt[:, 0] = 1
    for j, c in enumerate(cols):
        Xt[:, 1 + j] = df_test[c].to_numpy(dtype=np.float32)
    for i, k in enumerate(keys[0]):
        Xt[:, off[0] + i] = (y_test.to_numpy() == k)
    pred = Xt[:, feat] @ w
    r2_triple = r2_score(df_test[cols[0]], pred)

    tr = pd.get_dummies(df_train.assign(target=y_train), columns=["target"])
    te = pd.get_dummies(df_test.assign(target=y_test), columns=["target"])
    reg = LinearRegression().fit(tr.drop(columns=[cols[0]]), tr[cols[0]])
    r2_py = reg.score(te.drop(columns=[cols[0]]), te[cols[0]])
    assert round(r2_triple, 3) == round(r2_py, 3), (r2_triple, r2_py)
