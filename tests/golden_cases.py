"""The reference's ring-op known-answer tests, replayed against any backend.

Each case restates one test function of duckdb_extension/test/python/test_{sum,lift,mul,
nb_sum,nb_lift,nb_mul}.py as operations on the 5-row table; the expected values are the
literals those tests assert (tests/golden/ring_goldens.json).  A backend offers:

    sum_to(num_cols, cat_cols, nb)            -> blob      sum_to_triple_n_m / sum_to_nb_agg_n_m
    lift(num_cols, cat_cols, nb)              -> [blob]    to_cofactor / to_nb_agg (one per row)
    sum_lifted(blobs, nb)                     -> blob      sum_triple / sum_nb_agg
    multiply(blob_a, blob_b, nb)              -> blob      multiply_triple / multiply_nb_agg

GROUP BY is expressed by the caller: the rows of each group are handed over separately, which
is what the per-row state pointers of the reference's update amount to
(sum_no_lift.cpp:84,94,139).
"""
import numpy as np

from triple_fmt import blob_to_dict


def _expected(goldens, fname, test_index):
    t = goldens[fname]["tests"][test_index]
    exp = sorted(t["expected"], key=lambda e: e["row"])
    return t, [e["value"] for e in exp]


def cases(goldens, tables):
    """Yield (case_id, fn(backend) -> list of (got_dict, want_dict))."""
    out = []

    # ---- test_sum.py / test_nb_sum.py ------------------------------------------------------
    for fname, nb in (("test_sum.py", False), ("test_nb_sum.py", True)):
        T = tables[fname]
        names = "agg"

        def everything(be, T=T, nb=nb, fname=fname):
            _, want = _expected(goldens, fname, 0)
            got = be.sum_to(T.num("abc"), T.cat("def"), nb)
            return [(blob_to_dict(got, "agg"), want[0])]

        def group_by(be, T=T, nb=nb, fname=fname):
            _, want = _expected(goldens, fname, 1)
            res = []
            for gi, gb in enumerate((1, 2)):          # res[0] is gb=1 (N=2), res[1] is gb=2
                mk = T.mask(gb)
                got = be.sum_to(T.num("abc", mk), T.cat("def", mk), nb)
                res.append((blob_to_dict(got, "agg"), want[gi]))
            return res

        def having(be, T=T, nb=nb, fname=fname):
            _, want = _expected(goldens, fname, 2)
            mk = T.mask(2)
            got = be.sum_to(T.num("abc", mk), T.cat("def", mk), nb)
            return [(blob_to_dict(got, "agg"), want[0])]

        def fused_equals_unfused(be, T=T, nb=nb, fname=fname):
            # test_sum_group_by / test_sum_having: sum_to_triple == sum_triple(to_cofactor)
            res = []
            for gb in (1, 2):
                mk = T.mask(gb)
                fused = be.sum_to(T.num("abc", mk), T.cat("def", mk), nb)
                unfused = be.sum_lifted(be.lift(T.num("abc", mk), T.cat("def", mk), nb), nb)
                res.append((blob_to_dict(unfused, "agg"), blob_to_dict(fused, "agg")))
            return res

        out += [(fname + "::everything", everything), (fname + "::group_by", group_by),
                (fname + "::having", having), (fname + "::fused_equals_unfused", fused_equals_unfused)]

    # ---- test_lift.py / test_nb_lift.py ----------------------------------------------------
    for fname, nb in (("test_lift.py", False), ("test_nb_lift.py", True)):
        T = tables[fname]

        def lift_case(idx, num, cat, mask_gb=None, expr=False, T=T, nb=nb, fname=fname):
            def fn(be):
                _, want = _expected(goldens, fname, idx)
                mk = T.mask(mask_gb) if mask_gb is not None else None
                if expr:                               # to_cofactor(a+b+c): float column expr
                    ncols = [T.cols["a"] + T.cols["b"] + T.cols["c"]]
                    ccols = []
                else:
                    ncols, ccols = T.num(num, mk), T.cat(cat, mk)
                got = be.lift(ncols, ccols, nb)
                return [(blob_to_dict(g, "num"), w) for g, w in zip(got, want)]
            return fn

        out += [(fname + "::lift_all", lift_case(0, "abc", "def")),
                (fname + "::lift_single_int_column", lift_case(1, "", "e")),
                (fname + "::lift_single_float_column", lift_case(2, "a", "")),
                (fname + "::lift_with_where", lift_case(3, "abc", "def", mask_gb=2)),
                (fname + "::lift_with_sum_of_columns", lift_case(4, "", "", expr=True))]

    # ---- test_mul.py / test_nb_mul.py ------------------------------------------------------
    for fname, nb in (("test_mul.py", False), ("test_nb_mul.py", True)):
        T = tables[fname]

        def side(be, which, gb, T=T, nb=nb):
            mk = T.mask(gb)
            if which == "A":                           # sum_to_triple_2_2(b,c,d,e)
                return be.sum_to(T.num("bc", mk), T.cat("de", mk), nb)
            return be.sum_to(T.num("ac", mk), T.cat("df", mk), nb)   # (a,c,d,f)

        def mul_everything(be, nb=nb, fname=fname, side=side):
            _, want = _expected(goldens, fname, 0)
            got = be.multiply(side(be, "A", 1), side(be, "B", 2), nb)
            return [(blob_to_dict(got, "num"), want[0])]

        def mul_cross_join(be, nb=nb, fname=fname, side=side):
            # 2x2 cross join of the grouped triples.  Rows 0 and 3 of the reference's expected
            # output are the consistent pairings (g1 x g1, g2 x g2).  Rows 1 and 2 mix N from
            # one pairing with list payloads from another (an artefact of how the reference
            # indexes list children of cross-join dictionary vectors) and cannot be derived
            # from the inputs; they are not replayed.
            _, want = _expected(goldens, fname, 1)
            res = []
            for row, (ga, gb) in ((0, (1, 1)), (3, (2, 2))):
                got = be.multiply(side(be, "A", ga), side(be, "B", gb), nb)
                res.append((blob_to_dict(got, "num"), want[row]))
            return res

        out += [(fname + "::multiply_everything", mul_everything),
                (fname + "::multiply_cross_join", mul_cross_join)]
        if not nb:
            def mul_equi_join(be, nb=nb, fname=fname, side=side):
                _, want = _expected(goldens, fname, 2)
                res = []
                for row, g in ((0, 1), (1, 2)):
                    got = be.multiply(side(be, "A", g), side(be, "B", g), nb)
                    res.append((blob_to_dict(got, "num"), want[row]))
                return res
            out.append((fname + "::multiply_equi_join", mul_equi_join))
    return out
