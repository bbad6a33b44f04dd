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


def test_all_reference_literals_are_covered():
    """Every expected literal in the fixtures is replayed, except the two documented
    cross-join rows per multiply file (see golden_cases.mul_cross_join)."""
    total = sum(len(t["expected"]) for f in _G.values() for t in f["tests"])
    replayed = 0
    for _, fn in _CASES:
        replayed += len(fn(OracleBackend(orc.WIDE)))
    # the 2 fused_equals_unfused cases replay 2 equalities each without literals
    assert replayed - 4 == total - 4
