"""ctypes binding of the batched ring operations (include/cofactor_hip.h, cofactor_tvec and
cofactor_groups) for tests and tools: vectors of triples as torch (device) or numpy (host) arrays in
the reference's nested-vector layout, and the GROUP BY state pool."""
import ctypes as C

import numpy as np

from . import NB, TRIPLE, _check, _ptr_array, lib

_u64, _i32, _vp = C.c_uint64, C.c_int32, C.c_void_p

# (field, dtype) in the order of the C struct
_ARRAYS = [("N", np.int32), ("lin_e", np.uint64), ("lin", np.float32), ("quad_e", np.uint64), ("quad", np.float32),
           ("lc_outer", np.uint64), ("lc_sub", np.uint64), ("lc_key", np.int32), ("lc_val", np.float32),
           ("nc_outer", np.uint64), ("nc_sub", np.uint64), ("nc_key", np.int32), ("nc_val", np.float32),
           ("cc_outer", np.uint64), ("cc_sub", np.uint64), ("cc_key1", np.int32), ("cc_key2", np.int32),
           ("cc_val", np.float32)]


class TVecStruct(C.Structure):
    _fields_ = ([("count", _u64), ("n", _i32), ("m", _i32), ("kind", _i32)] + [(name, _vp) for name, _ in _ARRAYS] +
                [(f, _u64) for f in ("lc_cap", "nc_cap", "cc_cap", "lin_len", "quad_len", "lc_subs", "nc_subs", "cc_subs")])


def _tri(k):
    return k * (k + 1) // 2


def _bind():
    L = lib()
    if getattr(L, "_ring_bound", False):
        return L
    pp, pt = C.POINTER(C.c_void_p), C.POINTER(TVecStruct)
    pu = C.POINTER(_u64)
    L.cofactor_lift_device.argtypes = [_vp, pp, C.c_int, pp, C.c_int, _u64, C.c_int, pt]
    L.cofactor_lift_host_tvec.argtypes = [_vp, pp, C.c_int, pp, C.c_int, _u64, C.c_int, pt]
    L.cofactor_agg_update_tvec_device.argtypes = [_vp, pt]
    L.cofactor_agg_update_tvec_host.argtypes = [_vp, pt]
    L.cofactor_multiply_device.argtypes = [_vp, pt, _vp, pt, _vp, _u64, pt, pu, pu, pu]
    L.cofactor_multiply_host.argtypes = [_vp, pt, _vp, pt, _vp, _u64, pt, pu, pu, pu]
    L.cofactor_groups_create.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, pp]
    L.cofactor_groups_destroy.argtypes = [_vp]
    L.cofactor_groups_destroy.restype = None
    L.cofactor_groups_update_device.argtypes = [_vp, _vp, pp, pp, _u64]
    L.cofactor_groups_update_host.argtypes = [_vp, _vp, pp, pp, _u64]
    L.cofactor_groups_count.argtypes = [_vp, pu]
    L.cofactor_groups_combine.argtypes = [_vp, _i32, _i32]
    L.cofactor_groups_reset_group.argtypes = [_vp, _i32]
    L.cofactor_groups_finalize.argtypes = [_vp, _i32, _vp, _u64, pu]
    L.cofactor_groups_to_tvec.argtypes = [_vp, pt, _vp, pu, pu, pu]
    L._ring_bound = True
    return L


class TVec:
    """A vector of triples: arrays (torch tensors on the GPU, or numpy arrays) + the C struct."""

    def __init__(self, rows, n, m, kind, lc, nc, cc, device=None, regular=True):
        """Allocates every array for `rows` triples of shape (n, m, kind) whose payload arrays hold
        lc / nc / cc entries.  device=None: numpy (host); else torch tensors on that device."""
        self.rows, self.n, self.m, self.kind = rows, n, m, kind
        T = n if kind else _tri(n)
        nm, Tm = (0, 0) if kind else (n * m, _tri(m))
        sizes = {"N": rows, "lin_e": 2 * rows, "lin": rows * n, "quad_e": 2 * rows, "quad": rows * T,
                 "lc_outer": 2 * rows, "lc_sub": 2 * rows * m, "lc_key": lc, "lc_val": lc,
                 "nc_outer": 2 * rows, "nc_sub": 2 * rows * nm, "nc_key": nc, "nc_val": nc,
                 "cc_outer": 2 * rows, "cc_sub": 2 * rows * Tm, "cc_key1": cc, "cc_key2": cc, "cc_val": cc}
        self.a = {}
        if device is None:
            for name, dt in _ARRAYS:
                self.a[name] = np.zeros(max(1, sizes[name]), dtype=dt)
        else:
            import torch
            tdt = {np.int32: torch.int32, np.float32: torch.float32, np.uint64: torch.int64}
            for name, dt in _ARRAYS:
                self.a[name] = torch.zeros(max(1, sizes[name]), dtype=tdt[dt], device=device)
        self.device = device
        s = TVecStruct()
        s.count, s.n, s.m, s.kind = rows, n, m, kind
        for name, _ in _ARRAYS:
            arr = self.a[name]
            setattr(s, name, arr.ctypes.data if device is None else arr.data_ptr())
        s.lc_cap, s.nc_cap, s.cc_cap = lc, nc, cc
        s.lin_len, s.quad_len = rows * n, rows * T
        s.lc_subs, s.nc_subs, s.cc_subs = rows * m, rows * nm, rows * Tm
        self.struct = s

    def host(self):
        """name -> numpy array (copied off the device if need be)."""
        if self.device is None:
            return self.a
        out = {}
        for name, dt in _ARRAYS:
            h = self.a[name].cpu().numpy()
            out[name] = h.view(np.uint64) if dt is np.uint64 else h
        return out

    def to_blobs(self):
        """Every row as a flat triple blob (the header's format), for comparisons."""
        h = self.host()
        n, m, kind = self.n, self.m, self.kind
        s = self.struct
        T = n if kind else _tri(n)
        blobs = []
        for i in range(int(s.count)):
            b = [float(kind), float(n), float(m), float(h["N"][i])]
            lo, ln = int(h["lin_e"][2 * i]), int(h["lin_e"][2 * i + 1])
            assert ln == n
            b += [float(v) for v in h["lin"][lo:lo + ln]]
            qo, qn = int(h["quad_e"][2 * i]), int(h["quad_e"][2 * i + 1])
            assert qn == T
            b += [float(v) for v in h["quad"][qo:qo + qn]]

            def lists(outer, sub, keys, vals, expect):
                oo, on = int(h[outer][2 * i]), int(h[outer][2 * i + 1])
                assert on == expect, (outer, on, expect)
                for j in range(oo, oo + on):
                    eo, en = int(h[sub][2 * j]), int(h[sub][2 * j + 1])
                    b.append(float(en))
                    for e in range(eo, eo + en):
                        for k in keys:
                            b.append(float(h[k][e]))
                        b.append(float(h[vals][e]))
            if m:
                lists("lc_outer", "lc_sub", ["lc_key"], "lc_val", m)
                if not kind:
                    lists("nc_outer", "nc_sub", ["nc_key"], "nc_val", n * m)
                    lists("cc_outer", "cc_sub", ["cc_key1", "cc_key2"], "cc_val", _tri(m))
            blobs.append(np.array(b, dtype=np.float64))
        return blobs


def tvec_from_blobs(blobs, device=None):
    """Flat blobs (all of one shape) -> a vector of triples (host arrays, or copied to `device`)."""
    kind, n, m = int(blobs[0][0]), int(blobs[0][1]), int(blobs[0][2])
    T = n if kind else _tri(n)
    nm, Tm = (0, 0) if kind else (n * m, _tri(m))
    rows = len(blobs)
    N, lin, quad = [], [], []
    fam = [([], [], [], []) for _ in range(3)]       # (sub entries, key1, key2, val)
    for b in blobs:
        assert (int(b[0]), int(b[1]), int(b[2])) == (kind, n, m)
        N.append(int(b[3]))
        pos = 4
        lin += list(b[pos:pos + n]); pos += n
        quad += list(b[pos:pos + T]); pos += T
        for f, (count, width) in enumerate(((m, 2), (nm, 2), (Tm, 3))):
            if kind and f:
                continue
            sub, k1, k2, val = fam[f]
            for _ in range(count):
                ln = int(b[pos]); pos += 1
                sub += [len(val), ln]
                for _e in range(ln):
                    k1.append(int(b[pos]))
                    if width == 3:
                        k2.append(int(b[pos + 1]))
                    val.append(b[pos + width - 1])
                    pos += width
    t = TVec(rows, n, m, kind, len(fam[0][3]), len(fam[1][3]), len(fam[2][3]), device=None)
    a = t.a
    a["N"][:rows] = N
    a["lin"][:rows * n] = lin
    a["quad"][:rows * T] = quad
    for i in range(rows):
        a["lin_e"][2 * i:2 * i + 2] = (i * n, n)
        a["quad_e"][2 * i:2 * i + 2] = (i * T, T)
        a["lc_outer"][2 * i:2 * i + 2] = (i * m, m)
        a["nc_outer"][2 * i:2 * i + 2] = (i * nm, nm)
        a["cc_outer"][2 * i:2 * i + 2] = (i * Tm, Tm)
    for f, (sname, k1n, k2n, vn) in enumerate((("lc_sub", "lc_key", None, "lc_val"), ("nc_sub", "nc_key", None, "nc_val"),
                                                ("cc_sub", "cc_key1", "cc_key2", "cc_val"))):
        sub, k1, k2, val = fam[f]
        a[sname][:len(sub)] = sub
        a[k1n][:len(k1)] = k1
        if k2n:
            a[k2n][:len(k2)] = k2
        a[vn][:len(val)] = val
    if device is None:
        return t
    import torch
    d = TVec(rows, n, m, kind, t.struct.lc_cap, t.struct.nc_cap, t.struct.cc_cap, device=device)
    for name, dt in _ARRAYS:
        src = a[name].view(np.int64) if dt is np.uint64 else a[name]
        d.a[name].copy_(torch.from_numpy(src))
    torch.cuda.synchronize()
    return d


def lift_device(ctx, num, cat, kind=TRIPLE):
    """to_cofactor over device columns (torch tensors) -> device TVec."""
    L = _bind()
    n, m = len(num), len(cat)
    rows = (num + cat)[0].numel()
    out = TVec(rows, n, m, kind, rows * m, 0 if kind else rows * n * m, 0 if kind else rows * _tri(m), device=(num + cat)[0].device)
    import torch
    torch.cuda.synchronize()
    _check(L.cofactor_lift_device(ctx._h, _ptr_array([t.data_ptr() for t in num]), n,
                                  _ptr_array([t.data_ptr() for t in cat]), m, rows, kind, C.byref(out.struct)))
    ctx.synchronize()
    return out


def lift_host(ctx, num, cat, kind=TRIPLE):
    """to_cofactor over host columns (numpy) -> host TVec (through the GPU)."""
    L = _bind()
    num = [np.ascontiguousarray(c, dtype=np.float32) for c in num]
    cat = [np.ascontiguousarray(c, dtype=np.int32) for c in cat]
    n, m = len(num), len(cat)
    rows = len((num + cat)[0])
    out = TVec(rows, n, m, kind, rows * m, 0 if kind else rows * n * m, 0 if kind else rows * _tri(m))
    _check(L.cofactor_lift_host_tvec(ctx._h, _ptr_array([c.ctypes.data for c in num]), n,
                                     _ptr_array([c.ctypes.data for c in cat]), m, rows, kind, C.byref(out.struct)))
    return out


def update_tvec(agg, tv):
    """sum_triple: adds every row of the vector (device or host) to the aggregate."""
    L = _bind()
    fn = L.cofactor_agg_update_tvec_host if tv.device is None else L.cofactor_agg_update_tvec_device
    _check(fn(agg._h, C.byref(tv.struct)))


def multiply(ctx, a, b, a_sel=None, b_sel=None, rows=None):
    """multiply_triple over two vectors (both on the device, or both on the host), row i = a[a_sel[i]] x b[b_sel[i]]."""
    L = _bind()
    dev = a.device
    fn = L.cofactor_multiply_host if dev is None else L.cofactor_multiply_device
    if rows is None:
        rows = len(a_sel) if a_sel is not None else a.rows
    keep = []

    def selptr(s):
        if s is None:
            return None
        if dev is None:
            arr = np.ascontiguousarray(s, dtype=np.uint32)
            keep.append(arr)
            return arr.ctypes.data
        import torch
        if hasattr(s, "data_ptr"):                  # already a device tensor of 32-bit indices
            assert s.dtype == torch.int32 and s.is_contiguous()
            keep.append(s)
            return s.data_ptr()
        t = torch.as_tensor(np.ascontiguousarray(s, dtype=np.int64), device=dev).to(torch.int32)
        keep.append(t)
        torch.cuda.synchronize()
        return t.data_ptr()
    pa, pb = selptr(a_sel), selptr(b_sel)
    need = [_u64(0), _u64(0), _u64(0)]
    _check(fn(ctx._h, C.byref(a.struct), pa, C.byref(b.struct), pb, rows, None, C.byref(need[0]), C.byref(need[1]), C.byref(need[2])))
    out = TVec(rows, a.n + b.n, a.m + b.m, a.kind, need[0].value, need[1].value, need[2].value, device=dev)
    _check(fn(ctx._h, C.byref(a.struct), pa, C.byref(b.struct), pb, rows, C.byref(out.struct), None, None, None))
    if dev is not None:
        ctx.synchronize()          # the device form is asynchronous on the context stream
    return out


class Groups:
    """GROUP BY state pool (cofactor_groups)."""

    def __init__(self, ctx, n, m, kind=TRIPLE, is_key=True):
        L = _bind()
        h = C.c_void_p()
        _check(L.cofactor_groups_create(ctx._h, n, m, kind, int(is_key), C.byref(h)))
        self._h, self.ctx, self.n, self.m, self.kind = h, ctx, n, m, kind

    def close(self):
        if getattr(self, "_h", None):
            _bind().cofactor_groups_destroy(self._h)
            self._h = None

    def update_device(self, gid, num, cat):
        import torch
        torch.cuda.synchronize()
        _check(_bind().cofactor_groups_update_device(self._h, gid.data_ptr(), _ptr_array([t.data_ptr() for t in num]),
                                                     _ptr_array([t.data_ptr() for t in cat]), gid.numel()))

    def update_host(self, gid, num, cat):
        g = np.ascontiguousarray(gid, dtype=np.int32)
        num = [np.ascontiguousarray(c, dtype=np.float32) for c in num]
        cat = [np.ascontiguousarray(c, dtype=np.int32) for c in cat]
        _check(_bind().cofactor_groups_update_host(self._h, g.ctypes.data, _ptr_array([c.ctypes.data for c in num]),
                                                   _ptr_array([c.ctypes.data for c in cat]), g.size))

    def count(self):
        c = _u64(0)
        _check(_bind().cofactor_groups_count(self._h, C.byref(c)))
        return c.value

    def combine(self, dst, src):
        _check(_bind().cofactor_groups_combine(self._h, dst, src))

    def reset_group(self, gid):
        _check(_bind().cofactor_groups_reset_group(self._h, gid))

    def finalize(self, gid):
        L = _bind()
        need = _u64(0)
        _check(L.cofactor_groups_finalize(self._h, gid, None, 0, C.byref(need)))
        out = np.empty(need.value, dtype=np.float64)
        _check(L.cofactor_groups_finalize(self._h, gid, out.ctypes.data, out.size, C.byref(need)))
        return out

    def to_tvec(self, device):
        """-> (device TVec of all groups in ascending group order, int32 tensor of their keys)."""
        import torch
        L = _bind()
        need = [_u64(0), _u64(0), _u64(0)]
        _check(L.cofactor_groups_to_tvec(self._h, None, None, C.byref(need[0]), C.byref(need[1]), C.byref(need[2])))
        G = self.count()
        out = TVec(G, self.n, self.m, self.kind, need[0].value, need[1].value, need[2].value, device=device)
        keys = torch.zeros(max(1, G), dtype=torch.int32, device=device)
        torch.cuda.synchronize()
        _check(L.cofactor_groups_to_tvec(self._h, C.byref(out.struct), keys.data_ptr(), None, None, None))
        return out, keys[:G]
