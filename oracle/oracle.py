"""ctypes loader for the CPU oracle (oracle/cofactor_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (duckdb-imputation_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FAITHFUL, WIDE = 0, 1     # float/int32 accumulators (the reference) | double/int64 (truth)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libcofactor_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libcofactor_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_state_new.restype = C.c_void_p
        L.orc_state_new.argtypes = [C.c_int]
        L.orc_state_free.argtypes = [C.c_void_p]
        pp = C.POINTER(C.c_void_p)
        L.orc_update.argtypes = [pp, C.c_int, C.c_void_p, pp, C.c_int, pp, C.c_int,
                                 C.c_int64, C.c_int]
        L.orc_update_mt.argtypes = [C.c_void_p, pp, C.c_int, pp, C.c_int, C.c_int64, C.c_int,
                                    C.c_int]
        L.orc_update_mt_ex.argtypes = [C.c_void_p, pp, C.c_int, pp, C.c_int, C.c_int64, C.c_int,
                                       C.c_int, C.c_int]
        L.orc_combine.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_finalize.restype = C.c_int64
        L.orc_finalize.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_lift.restype = C.c_int64
        L.orc_lift.argtypes = [pp, C.c_int, pp, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                               C.c_void_p]
        L.orc_sum_blobs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_multiply.restype = C.c_int64
        L.orc_multiply.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_add_sub.restype = C.c_int64
        L.orc_add_sub.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _LIB = L
    return _LIB


def _colptrs(cols, dtype):
    cols = [np.ascontiguousarray(c, dtype=dtype) for c in cols]
    arr = (C.c_void_p * max(1, len(cols)))(*[c.ctypes.data for c in cols])
    return cols, arr


class State:
    """One aggregate state (the oracle's SumState)."""

    def __init__(self, mode=WIDE):
        self.mode = mode
        self._h = lib().orc_state_new(mode)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_state_free(self._h)
            self._h = None

    def update(self, num_cols, cat_cols, nb=False, threads=0, per_row_pointers=False):
        """sum_to_triple_n_m / sum_to_nb_agg_n_m over whole columns (2048-row chunks).
        per_row_pointers: every chunk through a vector of 2048 state pointers read per row, as
        DuckDB's executor drives the reference (bench.py's cpu_baseline)."""
        num, pn = _colptrs(num_cols, np.float32)
        cat, pc = _colptrs(cat_cols, np.int32)
        rows = len(num[0]) if num else (len(cat[0]) if cat else 0)
        if (threads and threads > 0) or per_row_pointers:
            lib().orc_update_mt_ex(self._h, pn, len(num), pc, len(cat), rows, int(nb), max(1, int(threads)),
                                   int(per_row_pointers))
        else:
            st = (C.c_void_p * 1)(self._h)
            lib().orc_update(st, 1, None, pn, len(num), pc, len(cat), rows, int(nb))
        return self

    def combine(self, other):
        assert lib().orc_combine(self._h, other._h) == 0
        return self

    def sum_blobs(self, blobs):
        """sum_triple / sum_nb_agg: add already-lifted triples (list of blobs)."""
        offs = np.zeros(len(blobs) + 1, dtype=np.int64)
        for i, b in enumerate(blobs):
            offs[i + 1] = offs[i] + len(b)
        flat = np.concatenate(blobs) if blobs else np.zeros(0)
        flat = np.ascontiguousarray(flat, dtype=np.float64)
        lib().orc_sum_blobs(self._h, flat.ctypes.data, offs.ctypes.data, len(blobs))
        return self

    def finalize(self):
        n = lib().orc_finalize(self._h, None)
        out = np.empty(n, dtype=np.float64)
        lib().orc_finalize(self._h, out.ctypes.data)
        return out


def grouped_update(num_cols, cat_cols, group, n_groups, nb=False, mode=WIDE):
    """GROUP BY: group[i] in [0, n_groups) plays the per-row state pointer."""
    num, pn = _colptrs(num_cols, np.float32)
    cat, pc = _colptrs(cat_cols, np.int32)
    g = np.ascontiguousarray(group, dtype=np.int32)
    states = [State(mode) for _ in range(n_groups)]
    arr = (C.c_void_p * n_groups)(*[s._h for s in states])
    lib().orc_update(arr, n_groups, g.ctypes.data, pn, len(num), pc, len(cat), len(g), int(nb))
    return states


def lift(num_cols, cat_cols, nb=False):
    """to_cofactor / to_nb_agg: one blob per row."""
    num, pn = _colptrs(num_cols, np.float32)
    cat, pc = _colptrs(cat_cols, np.int32)
    rows = len(num[0]) if num else (len(cat[0]) if cat else 0)
    offs = np.zeros(rows + 1, dtype=np.int64)
    n = lib().orc_lift(pn, len(num), pc, len(cat), rows, int(nb), None, offs.ctypes.data)
    out = np.empty(n, dtype=np.float64)
    lib().orc_lift(pn, len(num), pc, len(cat), rows, int(nb), out.ctypes.data, offs.ctypes.data)
    return [out[offs[i]:offs[i + 1]].copy() for i in range(rows)]


def _binary(fn, a, b, *args):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    n = fn(a.ctypes.data, b.ctypes.data, *args, None)
    out = np.empty(n, dtype=np.float64)
    fn(a.ctypes.data, b.ctypes.data, *args, out.ctypes.data)
    return out


def multiply(a, b, mode=FAITHFUL):
    return _binary(lib().orc_multiply, a, b, mode)


def add(a, b, mode=FAITHFUL):
    return _binary(lib().orc_add_sub, a, b, 0, mode)


def sub(a, b, mode=FAITHFUL):
    return _binary(lib().orc_add_sub, a, b, 1, mode)
