import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "duckdb-imputation_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def goldens():
    with open(os.path.join(ROOT, "tests", "golden", "ring_goldens.json")) as fh:
        return json.load(fh)


class RefTable:
    """The 5-row table of the reference's ring tests as columns (test_sum.py:15-16)."""

    def __init__(self, table):
        rows = table["rows"]
        self.cols = {}
        for j, (name, typ) in enumerate(zip(table["columns"], table["types"])):
            dt = np.float32 if typ == "FLOAT" else np.int32
            self.cols[name] = np.array([r[j] for r in rows], dtype=dt)
        self.n_rows = len(rows)

    def num(self, names, mask=None):
        return [self.cols[c][mask] if mask is not None else self.cols[c] for c in names]

    cat = num

    def mask(self, gb):
        return self.cols["gb"] == gb


@pytest.fixture(scope="session")
def ref_table(goldens):
    return {f: RefTable(v["table"]) for f, v in goldens.items()}
