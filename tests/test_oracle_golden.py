"""Pins the CPU oracle (oracle/cofactor_oracle.cpp) to every known-answer vector the
reference's own tests hold for the ring ops, in both oracle modes."""
import json
import os

import numpy as np
import pytest

from golden_cases import cases
from oracle import oracle as orc


class OracleBackend:
    def __init__(self, mode):
        self.mode = mode

    def sum_to(self, num, cat, nb):
        return orc.State(self.mode).update(num, cat, nb=nb).finalize()

    def lift(self, num, cat, nb):
        return orc.lift(num, cat, nb=nb)

    def sum_lifted(self, blobs, nb):
        return orc.State(self.mode).sum_blobs(blobs).finalize()

    def multiply(self, a, b, nb):
        return orc.multiply(a, b, self.mode)


def _load():
    root = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(root, "golden", "ring_goldens.json")) as fh:
        g = json.load(fh)
    from conftest import RefTable
    return g, {f: RefTable(v["table"]) for f, v in g.items()}


_G, _T = _load()
_CASES = cases(_G, _T)


@pytest.mark.parametrize("mode", [orc.FAITHFUL, orc.WIDE], ids=["faithful_f32", "wide_f64"])
@pytest.mark.parametrize("case", _CASES, ids=[c[0] for c in _CASES])
def test_oracle_matches_reference_goldens(case, mode):
    pairs = case[1](OracleBackend(mode))
    assert pairs
    for got, want in pairs:
        assert got == want          # exact ==, as the reference's tests do


# The reference literals that cannot be derived from their inputs: rows 1 and 2 of the cross-join
# GROUP BY case of each multiply test file (golden_cases.mul_cross_join says why).
EXCLUDED = {("test_mul.py", 1, 1), ("test_mul.py", 1, 2), ("test_nb_mul.py", 1, 1), ("test_nb_mul.py", 1, 2)}


def test_all_reference_literals_are_covered():
    """60 expected literals sit in the fixtures; 56 are replayed by the cases above and exactly
    the four named ones are not."""
    literals = {(fname, ti, e["row"]) for fname, f in _G.items() for ti, t in enumerate(f["tests"])
                for e in t["expected"]}
    assert len(literals) == 60
    assert EXCLUDED <= literals
    by_literal, self_checks = 0, 0
    for cid, fn in _CASES:
        pairs = fn(OracleBackend(orc.WIDE))
        if cid.endswith("::fused_equals_unfused"):
            self_checks += len(pairs)            # sum_to == sum(lift): two equalities, no literal
        else:
            by_literal += len(pairs)
    assert by_literal == 56 and by_literal == len(literals) - len(EXCLUDED)
    assert self_checks == 4
