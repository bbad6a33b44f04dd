"""Synthetic tables for bench.py and the tests: counter-based, so that any row
range of any column can be generated on its own and every sharding of a table is the SAME table.

    value(seed, stream, row) = splitmix64-style hash of the three

Numeric column k uses stream k, categorical column c stream 100 + c, the row-filter streams are
200+.  `uniform` gives float32 in [0, 1) (24 random bits), `integers` int32 in [0, K), `small_ints`
float32 whole numbers in [0, K) (exact-tier parity tables: every partial sum is exact).

Device side: plain torch integer ops on an arange chunk (measurement plumbing, nothing of the
product runs here).  Host side: the same arithmetic in numpy uint64, bit-identical.
"""
import numpy as np

_M1 = 0x9E3779B97F4A7C15
_M2 = 0xBF58476D1CE4E5B9
_M3 = 0x94D049BB133111EB
_MASK64 = (1 << 64) - 1
CHUNK = 1 << 26


def _s64(v):
    """Python int (mod 2^64) as the int64 with the same bits."""
    v &= _MASK64
    return v - (1 << 64) if v >= (1 << 63) else v


def _base(seed, stream):
    return (stream * _M2 + seed * _M3) & _MASK64


def hash_np(seed, stream, lo, hi):
    """uint64 hash of rows [lo, hi) on the host."""
    with np.errstate(over="ignore"):
        z = (np.arange(lo, hi, dtype=np.uint64) + np.uint64(1)) * np.uint64(_M1) + np.uint64(_base(seed, stream))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_M2)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_M3)
        return z ^ (z >> np.uint64(31))


def _hash_torch(torch, seed, stream, lo, hi, device):
    """int64 tensor holding the same 64 bits as hash_np (logical shifts emulated by masking)."""
    z = (torch.arange(lo, hi, dtype=torch.int64, device=device) + 1) * _s64(_M1) + _s64(_base(seed, stream))
    z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * _s64(_M2)
    z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * _s64(_M3)
    return z ^ ((z >> 31) & ((1 << 33) - 1))


def _column(torch, rows, dtype, device, stream):
    """An uninitialised device column.  COFACTOR_BENCH_STAGGER=<bytes> (multiple of 16; experiment
    knob) starts column `stream` that many bytes x (stream % 32) into its allocation, so that the
    same row of different columns does not sit at the same offset modulo 2 MiB."""
    import os
    stagger = int(os.environ.get("COFACTOR_BENCH_STAGGER", "0"))
    pad = (stagger * (stream % 32)) // 4
    if pad == 0:
        return torch.empty(rows, dtype=dtype, device=device)
    return torch.empty(rows + pad, dtype=dtype, device=device)[pad:]


def _fill(torch, out, seed, stream, lo, convert):
    rows = out.numel()
    for a in range(0, rows, CHUNK):
        b = min(rows, a + CHUNK)
        out[a:b] = convert(_hash_torch(torch, seed, stream, lo + a, lo + b, out.device))
    return out


def uniform(torch, seed, stream, lo, hi, device):
    """float32 uniform [0, 1): the top 24 bits of the hash times 2^-24."""
    out = _column(torch, hi - lo, torch.float32, device, stream)
    return _fill(torch, out, seed, stream, lo,
                 lambda z: ((z >> 40) & 0xFFFFFF).to(torch.float32) * (1.0 / (1 << 24)))


def integers(torch, seed, stream, lo, hi, K, device):
    """int32 uniform in [0, K) (K <= 2^24; the top 24 bits of the hash modulo K)."""
    out = _column(torch, hi - lo, torch.int32, device, stream)
    return _fill(torch, out, seed, stream, lo, lambda z: (((z >> 40) & 0xFFFFFF) % K).to(torch.int32))


def small_ints(torch, seed, stream, lo, hi, K, device):
    """float32 whole numbers in [0, K)."""
    out = _column(torch, hi - lo, torch.float32, device, stream)
    return _fill(torch, out, seed, stream, lo, lambda z: (((z >> 40) & 0xFFFFFF) % K).to(torch.float32))


def uniform_np(seed, stream, lo, hi):
    return ((hash_np(seed, stream, lo, hi) >> np.uint64(40)) & np.uint64(0xFFFFFF)).astype(np.float32) * np.float32(1.0 / (1 << 24))


def integers_np(seed, stream, lo, hi, K):
    return (((hash_np(seed, stream, lo, hi) >> np.uint64(40)) & np.uint64(0xFFFFFF)) % np.uint64(K)).astype(np.int32)


def small_ints_np(seed, stream, lo, hi, K):
    return integers_np(seed, stream, lo, hi, K).astype(np.float32)


def table(torch, seed, n, m, lo, hi, device, keys=16, exact=0):
    """Rows [lo, hi) of the bench table: n numeric columns (uniform, or whole numbers below `exact`
    when exact > 0) and m key columns with `keys` distinct keys."""
    if exact:
        num = [small_ints(torch, seed, k, lo, hi, exact, device) for k in range(n)]
    else:
        num = [uniform(torch, seed, k, lo, hi, device) for k in range(n)]
    cat = [integers(torch, seed, 100 + c, lo, hi, keys, device) for c in range(m)]
    return num, cat


def table_np(seed, n, m, lo, hi, keys=16, exact=0):
    if exact:
        num = [small_ints_np(seed, k, lo, hi, exact) for k in range(n)]
    else:
        num = [uniform_np(seed, k, lo, hi) for k in range(n)]
    cat = [integers_np(seed, 100 + c, lo, hi, keys) for c in range(m)]
    return num, cat
