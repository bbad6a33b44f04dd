"""CPU-side checks of the C ABI: the library loads, exports every symbol include/cofactor_hip.h
declares, refuses to run the aggregate without a GPU, and its host-only scalar ring ops
(lift / multiply / add / sub on flat blobs) reproduce the reference's golden vectors."""
import ctypes
import os
import re

import numpy as np
import pytest

import cofactor_hip
from oracle import oracle as orc
from triple_fmt import blob_to_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "cofactor_hip.h")).read()
    declared = set(re.findall(r"\b(cofactor_[a-z_]+)\s*\(", header))
    declared -= {"cofactor_status", "cofactor_kind"}
    assert declared == set(cofactor_hip.SYMBOLS), declared ^ set(cofactor_hip.SYMBOLS)
    lib = ctypes.CDLL(cofactor_hip.LIB_PATH)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.cofactor_abi_version() == 3


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(cofactor_hip.CofactorError) as e:
        cofactor_hip.Context(0)
    assert e.value.status == cofactor_hip.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_blob_len_walks_lists(goldens):
    want = goldens["test_sum.py"]["tests"][0]["expected"][0]["value"]
    from triple_fmt import dict_to_blob
    blob = dict_to_blob(want)
    assert cofactor_hip.blob_len(blob) == len(blob)
    assert cofactor_hip.blob_len(np.array([7.0, 1, 1, 1])) == 0          # bad kind


def test_host_ring_ops_match_goldens(goldens, ref_table):
    """to_cofactor / multiply_triple golden literals through the product's host ops."""
    for fname, kind in (("test_lift.py", cofactor_hip.TRIPLE), ("test_nb_lift.py", cofactor_hip.NB)):
        T = ref_table[fname]
        want = sorted(goldens[fname]["tests"][0]["expected"], key=lambda e: e["row"])
        got = cofactor_hip.lift_host(T.num("abc"), T.cat("def"), kind)
        assert [blob_to_dict(g, "num") for g in got] == [w["value"] for w in want]


def test_add_then_sub_round_trip(goldens):
    from triple_fmt import dict_to_blob
    exp = sorted(goldens["test_sum.py"]["tests"][1]["expected"], key=lambda e: e["row"])
    g1, g2 = dict_to_blob(exp[0]["value"]), dict_to_blob(exp[1]["value"])
    whole = dict_to_blob(goldens["test_sum.py"]["tests"][0]["expected"][0]["value"])
    # group 1 + group 2 == ungrouped (the reference's own literals on both sides)
    np.testing.assert_array_equal(cofactor_hip.add(g1, g2), whole)
    # and it agrees with the oracle's Value-level add (A12)
    np.testing.assert_array_equal(cofactor_hip.add(g1, g2), orc.add(g1, g2))
    # whole - group 2 has group 1's values on whole's key sets (zero counts stay, as sub.cpp keeps them)
    d = blob_to_dict(cofactor_hip.sub(whole, g2))
    assert d["N"] == 2 and d["lin_agg"] == blob_to_dict(g1)["lin_agg"]
    np.testing.assert_array_equal(cofactor_hip.sub(whole, g2), orc.sub(whole, g2))


def test_subtract_unknown_key_is_reported_not_fatal(goldens):
    from triple_fmt import dict_to_blob
    exp = sorted(goldens["test_sum.py"]["tests"][1]["expected"], key=lambda e: e["row"])
    g1, g2 = dict_to_blob(exp[0]["value"]), dict_to_blob(exp[1]["value"])
    out = cofactor_hip.sub(g1, g2)          # g2 has keys g1 lacks: reported and skipped
    assert b"not present in first triple" in cofactor_hip.lib().cofactor_last_error()
    np.testing.assert_array_equal(out, orc.sub(g1, g2))


def test_two_call_protocol_and_capacity_error(goldens):
    """out == NULL reports the size; a too-small buffer is COFACTOR_ERR_CAPACITY, never a write."""
    import ctypes as C
    from triple_fmt import dict_to_blob
    lib = cofactor_hip.lib()
    exp = sorted(goldens["test_sum.py"]["tests"][1]["expected"], key=lambda e: e["row"])
    a, b = dict_to_blob(exp[0]["value"]), dict_to_blob(exp[1]["value"])
    need = C.c_uint64(0)
    assert lib.cofactor_triple_add(a.ctypes.data, a.size, b.ctypes.data, b.size, None, 0, C.byref(need)) == cofactor_hip.OK
    assert need.value > 4
    small = np.full(4, -1.0)
    st = lib.cofactor_triple_add(a.ctypes.data, a.size, b.ctypes.data, b.size, small.ctypes.data, small.size, C.byref(need))
    assert st == cofactor_hip.ERR_CAPACITY and np.all(small == -1.0)
    assert b"too small" in lib.cofactor_last_error()


def test_malformed_blobs_are_rejected():
    lib = cofactor_hip.lib()
    bad_kind = np.array([2.0, 1, 0, 1, 0.5, 0.25])
    ok = np.array([0.0, 1, 0, 1, 0.5, 0.25])
    with pytest.raises(cofactor_hip.CofactorError) as e:
        cofactor_hip.add(bad_kind, ok)
    assert e.value.status == cofactor_hip.ERR_INVALID
    nb = np.array([1.0, 1, 0, 1, 0.5, 0.25])
    with pytest.raises(cofactor_hip.CofactorError):          # kinds differ
        cofactor_hip.multiply(ok, nb)
    with pytest.raises(cofactor_hip.CofactorError):
        cofactor_hip.add(ok, nb)


def test_truncated_and_lying_blobs_are_rejected_without_reading_past_them(goldens):
    """Every blob-taking entry point gets the blob's extent and checks each list header against
    it: a blob cut short, or one whose list header claims more entries than there are doubles, is
    COFACTOR_ERR_INVALID.  The blob sits at the very end of a buffer followed by a NaN canary page
    that a walk past the end would turn into a different answer."""
    from triple_fmt import dict_to_blob
    lib = cofactor_hip.lib()
    exp = sorted(goldens["test_sum.py"]["tests"][1]["expected"], key=lambda e: e["row"])
    full = dict_to_blob(exp[0]["value"])
    assert lib.cofactor_blob_len(full.ctypes.data, full.size) == full.size
    assert lib.cofactor_blob_len(full.ctypes.data, full.size + 100) == full.size
    for cut in (0, 3, 4, 7, full.size // 2, full.size - 1):
        assert lib.cofactor_blob_len(full.ctypes.data, cut) == 0, cut
        with pytest.raises(cofactor_hip.CofactorError) as e:
            cofactor_hip.add(full[:cut].copy(), full)
        assert e.value.status == cofactor_hip.ERR_INVALID
    n, m = int(full[1]), int(full[2])
    first_list = 4 + n + n * (n + 1) // 2
    lying = full.copy()
    lying[first_list] = 1e6                       # "a million keys" in a blob of a few hundred doubles
    assert lib.cofactor_blob_len(lying.ctypes.data, lying.size) == 0
    for fn in (cofactor_hip.add, cofactor_hip.sub, cofactor_hip.multiply):
        with pytest.raises(cofactor_hip.CofactorError):
            fn(lying, full)
        with pytest.raises(cofactor_hip.CofactorError):
            fn(full, lying)
    with pytest.raises(cofactor_hip.CofactorError):
        cofactor_hip.linreg_train(lying, 0)
    with pytest.raises(cofactor_hip.CofactorError):
        cofactor_hip.lda_train(full[: full.size - 2].copy(), 0)


def test_null_arguments_do_not_crash():
    import ctypes as C
    lib = cofactor_hip.lib()
    assert lib.cofactor_ctx_create(0, None) == cofactor_hip.ERR_INVALID
    assert lib.cofactor_agg_create(None, 1, 0, 0, None) == cofactor_hip.ERR_INVALID
    assert lib.cofactor_agg_finalize(None, None, 0, None) == cofactor_hip.ERR_INVALID
    assert lib.cofactor_agg_combine(None, None) == cofactor_hip.ERR_INVALID
    assert lib.cofactor_triple_multiply(None, 0, None, 0, None, 0, None) == cofactor_hip.ERR_INVALID
    lib.cofactor_agg_destroy(None)          # no-ops
    lib.cofactor_ctx_destroy(None)
    assert lib.cofactor_blob_len(None, 0) == 0
    assert lib.cofactor_dense_len(20, 0) == 1 + 20 + 210 and lib.cofactor_dense_len(20, 1) == 41


def test_multiply_matches_oracle_on_random_triples():
    """multiply_triple beyond the 5-row goldens: random shapes, integer values (exact in float)."""
    rng = np.random.default_rng(6)
    for trial in range(20):
        n1, m1, n2, m2 = (int(x) for x in rng.integers(0, 4, 4))
        if n1 + m1 == 0: n1 = 1
        if n2 + m2 == 0: m2 = 1
        def table(n, m, rows):
            return ([rng.integers(0, 5, rows).astype(np.float32) for _ in range(n)],
                    [rng.integers(-2, 4, rows).astype(np.int32) for _ in range(m)])
        for nb in (False, True):
            A = orc.State(orc.FAITHFUL).update(*table(n1, m1, 7), nb=nb).finalize()
            B = orc.State(orc.FAITHFUL).update(*table(n2, m2, 5), nb=nb).finalize()
            np.testing.assert_array_equal(cofactor_hip.multiply(A, B), orc.multiply(A, B, orc.FAITHFUL))
            if (n1, m1) == (n2, m2):
                np.testing.assert_array_equal(cofactor_hip.add(A, B), orc.add(A, B, orc.FAITHFUL))


def test_triple_text_round_trip(goldens):
    """N4: the triple as DuckDB's STRUCT literal text (imputation_base.cpp:46-49,116) and back.  Every
    reference literal: blob -> text -> blob is the identity; the text parses when written the way
    Python / DuckDB print it (repr of the golden dict); floats that need 17 digits keep them."""
    from triple_fmt import dict_to_blob
    count = 0
    for fname, doc in goldens.items():
        agg_names = "sum" in fname
        for t in doc["tests"]:
            for e in t["expected"]:
                blob = dict_to_blob(e["value"])
                text = cofactor_hip.to_text(blob, aggregate_names=agg_names)
                assert ("'lin_agg'" in text) == agg_names
                np.testing.assert_array_equal(cofactor_hip.from_text(text), blob)
                np.testing.assert_array_equal(cofactor_hip.from_text(repr(e["value"])), blob)
                count += 1
    assert count == 60
    rng = np.random.default_rng(3)
    from oracle import oracle as orc
    num = [rng.normal(size=500).astype(np.float32) * 1e-3 for _ in range(3)]
    cat = [rng.integers(-2_000_000_000, 2_000_000_000, 500).astype(np.int32) for _ in range(2)]
    for nb in (False, True):
        blob = orc.State(orc.WIDE).update(num, cat, nb=nb).finalize()      # doubles that are no floats
        text = cofactor_hip.to_text(blob)
        np.testing.assert_array_equal(cofactor_hip.from_text(text), blob)
        assert cofactor_hip.to_text(cofactor_hip.from_text(text)) == text
    nonfinite = np.array([0.0, 1, 0, 2, np.inf, -np.inf if False else np.nan])
    back = cofactor_hip.from_text(cofactor_hip.to_text(nonfinite))
    assert np.isinf(back[4]) and np.isnan(back[5])
    for bad in ("", "{", "{'N': 1}", "{'N': 1, 'lin_agg': [1.0], 'quad_agg': [1.0, 2.0], 'lin_cat': []}",
                "{'N': 1, 'lin_agg': [], 'quad_agg': [], 'lin_cat': [], 'quad_cat': []}", "{'N': x}"):
        with pytest.raises(cofactor_hip.CofactorError) as e:
            cofactor_hip.from_text(bad)
        assert e.value.status == cofactor_hip.ERR_INVALID


def test_flat_add_sub_of_equally_shaped_triples_equals_the_general_path():
    """Triples with identical ascending key lists (the cofactor of a table and of some of its rows, as
    the partitioned MICE loop adds and subtracts them) take a path that works on the flat blobs; the
    same triples with one key list reversed go through the general (map-based) path, which sorts: both
    must give the same blob.  Differently shaped triples still work."""
    rng = np.random.default_rng(12)
    rows = 4000
    num = [rng.integers(0, 9, rows).astype(np.float32) for _ in range(3)]
    cat = [rng.integers(0, 5, rows).astype(np.int32) for _ in range(3)]
    a = orc.State(orc.WIDE).update(num, cat).finalize()
    b = orc.State(orc.WIDE).update([c[:900] for c in num], [c[:900] for c in cat]).finalize()
    assert len(a) == len(b)

    def reversed_first_list(blob):
        out = np.array(blob, dtype=np.float64)
        n, m = int(out[1]), int(out[2])
        p = 4 + n + n * (n + 1) // 2
        ln = int(out[p])
        pairs = out[p + 1:p + 1 + 2 * ln].reshape(ln, 2)[::-1].copy()
        out[p + 1:p + 1 + 2 * ln] = pairs.reshape(-1)
        return out

    for fn in (cofactor_hip.add, cofactor_hip.sub):
        flat = fn(a, b)
        general = fn(reversed_first_list(a), reversed_first_list(b))
        assert np.array_equal(flat, general)
    d = blob_to_dict(cofactor_hip.sub(a, b))
    want = blob_to_dict(orc.State(orc.WIDE).update([c[900:] for c in num], [c[900:] for c in cat]).finalize())
    assert d["N"] == want["N"] and d["lin_agg"] == want["lin_agg"] and d["quad_agg"] == want["quad_agg"]
    assert [[e["key"] for e in l] for l in d["lin_cat"]] == [[e["key"] for e in l] for l in blob_to_dict(a)["lin_cat"]]
    small = orc.State(orc.WIDE).update([c[:7] for c in num], [c[:7] for c in cat]).finalize()   # fewer keys: general path
    assert blob_to_dict(cofactor_hip.add(cofactor_hip.sub(a, small), small))["N"] == rows
