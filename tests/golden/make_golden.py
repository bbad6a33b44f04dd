#!/usr/bin/env python3
"""Extract the known-answer vectors of the reference's own tests into JSON fixtures.

Reads (as text, never imports or executes) the six ring-op pytest files under
/root/reference/duckdb_extension/test/python/ and writes tests/golden/ring_goldens.json:
for every test function the SQL strings it sends (inputs: the 5-row table and the query)
and every expected nested-triple literal it asserts (outputs), keyed by result row.

Only DATA is extracted (table rows, query strings, expected values); no reference source
text is stored.  Run here (the reference is not present on the GPU box):
    python tests/golden/make_golden.py
"""
import ast
import json
import os
import re
import sys

REF = "/root/reference/duckdb_extension/test/python"
FILES = ["test_sum.py", "test_lift.py", "test_mul.py",
         "test_nb_sum.py", "test_nb_lift.py", "test_nb_mul.py"]
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ring_goldens.json")


def string_args(call):
    """Concatenate the (possibly implicitly joined) string literal of call's first arg."""
    if not call.args:
        return None
    a = call.args[0]
    if isinstance(a, ast.Constant) and isinstance(a.value, str):
        return a.value
    return None


def parse_insert(sql):
    rows = re.findall(r"\(([^()]*)\)", sql.split("VALUES", 1)[1])
    return [[float(x) if "." in x else int(x) for x in r.split(",")] for r in rows]


def extract(path):
    tree = ast.parse(open(path).read())
    out = {"table": None, "tests": []}
    for node in tree.body:
        if not isinstance(node, ast.FunctionDef):
            continue
        queries, expected = [], []
        for sub in ast.walk(node):
            if isinstance(sub, ast.Call) and isinstance(sub.func, ast.Attribute) \
                    and sub.func.attr == "execute":
                s = string_args(sub)
                if s is None:
                    continue
                if s.startswith("INSERT INTO test"):
                    out["table"] = {"columns": ["gb", "a", "b", "c", "d", "e", "f"],
                                    "types": ["INTEGER", "FLOAT", "FLOAT", "FLOAT",
                                              "INTEGER", "INTEGER", "INTEGER"],
                                    "rows": parse_insert(s)}
                elif s.upper().lstrip().startswith("SELECT"):
                    queries.append(s)
            # assert(res[i][0] == eval("<literal>"))
            if isinstance(sub, ast.Compare) and isinstance(sub.left, ast.Subscript):
                rhs = sub.comparators[0]
                if isinstance(rhs, ast.Call) and getattr(rhs.func, "id", "") == "eval":
                    lit = ast.literal_eval(string_args(rhs))
                    row = sub.left.value.slice
                    row = row.value if isinstance(row, ast.Constant) else None
                    expected.append({"row": row, "value": lit})
        if node.name == "duckdb_conn":
            continue
        out["tests"].append({"name": node.name, "queries": queries, "expected": expected})
    return out


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tests not found at " + REF)
    blob = {f: extract(os.path.join(REF, f)) for f in FILES}
    with open(OUT, "w") as fh:
        json.dump(blob, fh, indent=1, sort_keys=True)
    n = sum(len(t["expected"]) for f in blob.values() for t in f["tests"])
    print("wrote", OUT, "with", n, "expected triples")


if __name__ == "__main__":
    main()
