"""ctypes binding of libcofactor_hip.so (include/cofactor_hip.h) for tests and bench.py.

Thin by design: every call goes straight through the C ABI; torch is only used by callers for
device memory (the binding takes raw device pointers via ``tensor.data_ptr()``).  There is no
fallback of any kind: if the library is missing or there is no GPU, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COFACTOR_LIB", os.path.join(_HERE, "libcofactor_hip.so"))

TRIPLE, NB = 0, 1
OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_UNSUPPORTED, ERR_INTERNAL = range(7)

# every symbol include/cofactor_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "cofactor_last_error", "cofactor_abi_version", "cofactor_device_count",
    "cofactor_ctx_create", "cofactor_ctx_destroy", "cofactor_ctx_synchronize", "cofactor_ctx_stream",
    "cofactor_ctx_profile_enable", "cofactor_ctx_profile_kernel", "cofactor_ctx_profile_read", "cofactor_ctx_calibrate",
    "cofactor_agg_create", "cofactor_agg_destroy", "cofactor_agg_reset",
    "cofactor_agg_update_device", "cofactor_agg_update_device_masked", "cofactor_agg_update_host",
    "cofactor_agg_update_triples",
    "cofactor_agg_combine", "cofactor_agg_finalize",
    "cofactor_dense_len", "cofactor_agg_export_dense_device", "cofactor_agg_import_dense_device",
    "cofactor_agg_keys", "cofactor_agg_dict_signature", "cofactor_agg_align_keys",
    "cofactor_agg_tables_len", "cofactor_agg_export_tables_device", "cofactor_agg_import_tables_device",
    "cofactor_agg_sparse_lens", "cofactor_agg_sparse_is_list", "cofactor_agg_sparse_export_device",
    "cofactor_agg_sparse_assign_device",
    "cofactor_comm_unique_id", "cofactor_comm_create", "cofactor_comm_destroy", "cofactor_agg_allreduce",
    "cofactor_lift_host", "cofactor_triple_multiply", "cofactor_triple_add", "cofactor_triple_sub",
    "cofactor_lift_device", "cofactor_agg_update_tvec_device", "cofactor_multiply_device",
    "cofactor_lift_host_tvec", "cofactor_agg_update_tvec_host", "cofactor_multiply_host",
    "cofactor_groups_create", "cofactor_groups_destroy", "cofactor_groups_update_device",
    "cofactor_groups_update_host", "cofactor_groups_count", "cofactor_groups_combine", "cofactor_groups_reset_group", "cofactor_linreg_predict_rows_device",
    "cofactor_groups_finalize", "cofactor_groups_to_tvec",
    "cofactor_blob_len", "cofactor_triple_to_text", "cofactor_triple_from_text",
    "cofactor_linreg_train", "cofactor_lda_train",
    "cofactor_linreg_predict_device", "cofactor_lda_predict_device",
    "cofactor_linreg_predict_host", "cofactor_lda_predict_host",
]

_lib = None


class CofactorError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("cofactor status %d: %s" % (status, msg))
        self.status = status


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CofactorError(-1, "libcofactor_hip.so not built (run __graft_entry__.build())")
        # torch ships its own libamdhip64.so.7; two HIP runtimes in one process cannot both open
        # the GPU.  Loading torch first makes the loader resolve our NEEDED libamdhip64.so.7 to
        # the copy torch already mapped.  (C / C++ users link the system ROCm runtime as usual.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, pp, u64 = C.c_void_p, C.POINTER(C.c_void_p), C.c_uint64
        pu64 = C.POINTER(C.c_uint64)
        L.cofactor_last_error.restype = C.c_char_p
        L.cofactor_ctx_create.argtypes = [C.c_int, pp]
        L.cofactor_ctx_destroy.argtypes = [vp]
        L.cofactor_ctx_destroy.restype = None
        L.cofactor_ctx_synchronize.argtypes = [vp]
        L.cofactor_ctx_stream.argtypes = [vp]
        L.cofactor_ctx_stream.restype = vp
        L.cofactor_ctx_profile_enable.argtypes = [vp, C.c_int]
        L.cofactor_ctx_profile_kernel.argtypes = [vp]
        L.cofactor_ctx_profile_kernel.restype = C.c_char_p
        L.cofactor_ctx_profile_read.argtypes = [vp] + [C.POINTER(C.c_double), pu64] * 3
        L.cofactor_ctx_calibrate.argtypes = [vp, u64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.cofactor_agg_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, pp]
        L.cofactor_agg_destroy.argtypes = [vp]
        L.cofactor_agg_destroy.restype = None
        L.cofactor_agg_reset.argtypes = [vp]
        L.cofactor_agg_update_device.argtypes = [vp, pp, pp, u64]
        L.cofactor_agg_update_device_masked.argtypes = [vp, pp, pp, vp, u64]
        L.cofactor_agg_update_host.argtypes = [vp, pp, pp, pp, pp, vp, u64]
        L.cofactor_agg_update_triples.argtypes = [vp, vp, vp, u64]
        L.cofactor_agg_combine.argtypes = [vp, vp]
        L.cofactor_agg_finalize.argtypes = [vp, vp, u64, pu64]
        L.cofactor_dense_len.argtypes = [C.c_int, C.c_int]
        L.cofactor_dense_len.restype = u64
        L.cofactor_agg_export_dense_device.argtypes = [vp, vp]
        L.cofactor_agg_import_dense_device.argtypes = [vp, vp]
        L.cofactor_agg_keys.argtypes = [vp, vp, u64, pu64, vp]
        L.cofactor_agg_dict_signature.argtypes = [vp, pu64]
        L.cofactor_agg_align_keys.argtypes = [vp, vp, vp]
        L.cofactor_agg_tables_len.argtypes = [vp]
        L.cofactor_agg_tables_len.restype = u64
        L.cofactor_agg_export_tables_device.argtypes = [vp, vp]
        L.cofactor_agg_import_tables_device.argtypes = [vp, vp]
        L.cofactor_agg_sparse_lens.argtypes = [vp, pu64, u64]
        L.cofactor_agg_sparse_is_list.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32)]
        L.cofactor_agg_sparse_export_device.argtypes = [vp, C.c_int32, vp, vp]
        L.cofactor_agg_sparse_assign_device.argtypes = [vp, C.c_int32, vp, vp, u64]
        L.cofactor_comm_unique_id.argtypes = [vp]
        L.cofactor_comm_create.argtypes = [vp, vp, C.c_int, C.c_int, pp]
        L.cofactor_comm_destroy.argtypes = [vp]
        L.cofactor_comm_destroy.restype = None
        L.cofactor_agg_allreduce.argtypes = [vp, vp]
        L.cofactor_lift_host.argtypes = [pp, C.c_int, pp, C.c_int, u64, C.c_int, vp, u64, pu64, vp]
        for f in ("cofactor_triple_multiply", "cofactor_triple_add", "cofactor_triple_sub"):
            getattr(L, f).argtypes = [vp, u64, vp, u64, vp, u64, pu64]
        L.cofactor_blob_len.argtypes = [vp, u64]
        L.cofactor_blob_len.restype = u64
        L.cofactor_triple_to_text.argtypes = [vp, u64, C.c_int32, vp, u64, pu64]
        L.cofactor_triple_from_text.argtypes = [C.c_char_p, u64, vp, u64, pu64]
        i32, f32 = C.c_int32, C.c_float
        L.cofactor_linreg_train.argtypes = [vp, u64, i32, f32, f32, i32, i32, i32, vp, u64, pu64]
        L.cofactor_lda_train.argtypes = [vp, u64, i32, f32, i32, vp, u64, pu64]
        L.cofactor_linreg_predict_device.argtypes = [vp, vp, u64, i32, i32, u64, pp, i32, pp, i32, vp, u64, vp]
        L.cofactor_linreg_predict_rows_device.argtypes = [vp, vp, u64, i32, i32, u64, pp, i32, pp, i32, vp, vp, u64, vp]
        L.cofactor_lda_predict_device.argtypes = [vp, vp, u64, i32, i32, pp, i32, pp, i32, vp, u64, vp]
        L.cofactor_linreg_predict_host.argtypes = [vp, vp, u64, i32, i32, u64, pp, i32, pp, i32, u64, vp]
        L.cofactor_lda_predict_host.argtypes = [vp, vp, u64, i32, i32, pp, i32, pp, i32, u64, vp]
        _lib = L
    return _lib


def _check(status):
    if status != OK:
        raise CofactorError(status, lib().cofactor_last_error().decode())


def _ptr_array(ptrs):
    return (C.c_void_p * max(1, len(ptrs)))(*ptrs)


def _two_call(fn, *args):
    """The header's two-call protocol: size query, then fill."""
    need = C.c_uint64(0)
    _check(fn(*args, None, 0, C.byref(need)))
    out = np.empty(need.value, dtype=np.float64)
    _check(fn(*args, out.ctypes.data, out.size, C.byref(need)))
    return out


class Context:
    """One GPU (cofactor_ctx)."""

    def __init__(self, device=0):
        h = C.c_void_p()
        _check(lib().cofactor_ctx_create(device, C.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            lib().cofactor_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        if lib is not None:          # module globals are already torn down at interpreter exit
            self.close()

    def synchronize(self):
        _check(lib().cofactor_ctx_synchronize(self._h))

    @property
    def stream(self):
        """Raw hipStream_t of the context (an int)."""
        return lib().cofactor_ctx_stream(self._h)

    def profile(self, on=True):
        _check(lib().cofactor_ctx_profile_enable(self._h, int(on)))

    def profile_read(self):
        """-> dict({gram,cat,fused}_ms, {gram,cat,fused}_launches) since the previous read."""
        ms = [C.c_double(0) for _ in range(3)]
        ln = [C.c_uint64(0) for _ in range(3)]
        _check(lib().cofactor_ctx_profile_read(self._h, C.byref(ms[0]), C.byref(ln[0]), C.byref(ms[1]),
                                               C.byref(ln[1]), C.byref(ms[2]), C.byref(ln[2])))
        out = {}
        for name, a, b in zip(("gram", "cat", "fused"), ms, ln):
            out[name + "_ms"], out[name + "_launches"] = a.value, b.value
        out["fused_kernel"] = (lib().cofactor_ctx_profile_kernel(self._h) or b"").decode() or "fused_kernel"
        return out

    def calibrate(self, nbytes=4 << 30, reps=5):
        """-> (copy GB/s, read GB/s) of a plain float4 streaming kernel on this GPU."""
        c, r = C.c_double(0), C.c_double(0)
        _check(lib().cofactor_ctx_calibrate(self._h, nbytes, reps, C.byref(c), C.byref(r)))
        return c.value, r.value

    def aggregate(self, n_num, n_cat, kind=TRIPLE):
        return Aggregate(self, n_num, n_cat, kind)

    # ---- per-row predictors over device columns (torch tensors) or host columns (numpy) ----
    def linreg_predict(self, params, num_cols, cat_cols, out=None, mask=None, noise=False,
                       normalize=False, seed=0, row_ids=None):
        """linreg_predict(params, noise, normalize, feature columns.., key columns..): device
        tensors write `out` (float32 device tensor; only rows with mask != 0 when a mask is
        given), numpy columns return a new float32 array."""
        prm = np.ascontiguousarray(params, dtype=np.float32)
        if _on_device(num_cols, cat_cols):
            rows = out.numel()
            _check_cols(num_cols, cat_cols, rows, mask)
            assert str(out.dtype) == "torch.float32" and out.is_cuda and out.is_contiguous()
            _torch_handover([out])
            if row_ids is not None:                  # (uint32 device tensor: the rows' original places)
                assert row_ids.is_cuda and row_ids.is_contiguous() and row_ids.numel() == rows and row_ids.element_size() == 4
                _check(lib().cofactor_linreg_predict_rows_device(
                    self._h, prm.ctypes.data, prm.size, int(noise), int(normalize), seed,
                    _ptr_array([t.data_ptr() for t in num_cols]), len(num_cols),
                    _ptr_array([t.data_ptr() for t in cat_cols]), len(cat_cols),
                    None if mask is None else mask.data_ptr(), row_ids.data_ptr(), rows, out.data_ptr()))
                return out
            _check(lib().cofactor_linreg_predict_device(
                self._h, prm.ctypes.data, prm.size, int(noise), int(normalize), seed,
                _ptr_array([t.data_ptr() for t in num_cols]), len(num_cols),
                _ptr_array([t.data_ptr() for t in cat_cols]), len(cat_cols),
                None if mask is None else mask.data_ptr(), rows, out.data_ptr()))
            return out
        num = [np.ascontiguousarray(c, dtype=np.float32) for c in num_cols]
        cat = [np.ascontiguousarray(c, dtype=np.int32) for c in cat_cols]
        rows = len((num + cat)[0])
        res = np.empty(rows, dtype=np.float32)
        _check(lib().cofactor_linreg_predict_host(
            self._h, prm.ctypes.data, prm.size, int(noise), int(normalize), seed,
            _ptr_array([c.ctypes.data for c in num]), len(num),
            _ptr_array([c.ctypes.data for c in cat]), len(cat), rows, res.ctypes.data))
        return res

    def lda_predict(self, params, num_cols, cat_cols, out=None, mask=None, normalize=False,
                    emit_label=False):
        """lda_predict(params, normalize, feature columns.., key columns..) -> class index
        (the reference's return value) or, with emit_label, the class key."""
        prm = np.ascontiguousarray(params, dtype=np.float32)
        if _on_device(num_cols, cat_cols):
            rows = out.numel()
            _check_cols(num_cols, cat_cols, rows, mask)
            assert str(out.dtype) == "torch.int32" and out.is_cuda and out.is_contiguous()
            _torch_handover([out])
            _check(lib().cofactor_lda_predict_device(
                self._h, prm.ctypes.data, prm.size, int(normalize), int(emit_label),
                _ptr_array([t.data_ptr() for t in num_cols]), len(num_cols),
                _ptr_array([t.data_ptr() for t in cat_cols]), len(cat_cols),
                None if mask is None else mask.data_ptr(), rows, out.data_ptr()))
            return out
        num = [np.ascontiguousarray(c, dtype=np.float32) for c in num_cols]
        cat = [np.ascontiguousarray(c, dtype=np.int32) for c in cat_cols]
        rows = len((num + cat)[0])
        res = np.empty(rows, dtype=np.int32)
        _check(lib().cofactor_lda_predict_host(
            self._h, prm.ctypes.data, prm.size, int(normalize), int(emit_label),
            _ptr_array([c.ctypes.data for c in num]), len(num),
            _ptr_array([c.ctypes.data for c in cat]), len(cat), rows, res.ctypes.data))
        return res


def _torch_handover(tensors):
    """The library runs on its own stream: wait for what torch has queued on its current stream
    (fills, copies, generators writing these tensors) before a kernel of ours reads them."""
    for t in tensors:
        if t is not None and hasattr(t, "is_cuda") and t.is_cuda:
            import torch
            torch.cuda.current_stream(t.device).synchronize()
            return


def _on_device(num_cols, cat_cols):
    cols = list(num_cols) + list(cat_cols)
    return bool(cols) and hasattr(cols[0], "data_ptr")


def _check_cols(num_cols, cat_cols, rows, mask):
    for t, want in [(t, "torch.float32") for t in num_cols] + [(t, "torch.int32") for t in cat_cols]:
        assert str(t.dtype) == want and t.is_cuda and t.is_contiguous() and t.numel() == rows, \
            "columns must be 1-D contiguous device tensors of float32 / int32 with one entry per row"
    if mask is not None:
        assert str(mask.dtype) == "torch.uint8" and mask.is_cuda and mask.is_contiguous() \
            and mask.numel() == rows


class Aggregate:
    """One aggregate state (cofactor_agg): what a Triple::SumState is in the reference."""

    def __init__(self, ctx, n_num, n_cat, kind=TRIPLE):
        h = C.c_void_p()
        _check(lib().cofactor_agg_create(ctx._h, n_num, n_cat, kind, C.byref(h)))
        self._h, self.ctx, self.n, self.m, self.kind = h, ctx, n_num, n_cat, kind

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            lib().cofactor_agg_destroy(self._h)
        self._h = None

    def __del__(self):
        if lib is not None:
            self.close()

    def reset(self):
        _check(lib().cofactor_agg_reset(self._h))

    def update_device_ptrs(self, num_ptrs, cat_ptrs, rows):
        """Raw device pointers (ints), e.g. tensor.data_ptr()."""
        assert len(num_ptrs) == self.n and len(cat_ptrs) == self.m
        _check(lib().cofactor_agg_update_device(self._h, _ptr_array(num_ptrs), _ptr_array(cat_ptrs),
                                                rows))

    def update_device(self, num_tensors, cat_tensors):
        """Columns as 1-D contiguous torch tensors on this context's GPU (float32 / int32).
        Waits for torch's current stream first (update_device_ptrs does not: raw pointers are
        the caller's responsibility)."""
        _torch_handover(list(num_tensors) + list(cat_tensors))
        rows = None
        for t, want in [(t, "torch.float32") for t in num_tensors] + \
                       [(t, "torch.int32") for t in cat_tensors]:
            assert str(t.dtype) == want and t.is_contiguous() and t.dim() == 1 and t.is_cuda, \
                "columns must be 1-D contiguous device tensors of float32 / int32"
            rows = t.numel() if rows is None else rows
            assert t.numel() == rows
        self.update_device_ptrs([t.data_ptr() for t in num_tensors],
                                [t.data_ptr() for t in cat_tensors], rows or 0)

    def update_device_masked(self, num_tensors, cat_tensors, mask):
        """Like update_device, keeping only rows whose byte in `mask` (uint8 device tensor) is non-zero."""
        assert str(mask.dtype) == "torch.uint8" and mask.is_cuda and mask.is_contiguous()
        _torch_handover([mask])
        rows = mask.numel()
        for t in list(num_tensors) + list(cat_tensors):
            assert t.numel() == rows and t.is_cuda and t.is_contiguous()
        _check(lib().cofactor_agg_update_device_masked(
            self._h, _ptr_array([t.data_ptr() for t in num_tensors]),
            _ptr_array([t.data_ptr() for t in cat_tensors]), mask.data_ptr(), rows))

    def update_host(self, num_cols, cat_cols, num_sel=None, cat_sel=None, row_idx=None, rows=None):
        """One DataChunk of host columns (numpy), optional per-column selection vectors and the
        list of chunk rows that belong to this state (see the header)."""
        num = [np.ascontiguousarray(c, dtype=np.float32) for c in num_cols]
        cat = [np.ascontiguousarray(c, dtype=np.int32) for c in cat_cols]
        assert len(num) == self.n and len(cat) == self.m
        keep = [num, cat]

        def sel_array(sels, k):
            if sels is None:
                return None
            arrs = [None if s is None else np.ascontiguousarray(s, dtype=np.uint32) for s in sels]
            keep.append(arrs)
            return _ptr_array([0 if a is None else a.ctypes.data for a in arrs]) if k else None

        ns, cs = sel_array(num_sel, self.n), sel_array(cat_sel, self.m)
        ri = None if row_idx is None else np.ascontiguousarray(row_idx, dtype=np.uint32)
        if rows is None:
            if ri is not None:
                rows = len(ri)
            else:
                first = (num + cat)[0]
                firstsel = (list(num_sel or []) + list(cat_sel or []))
                rows = len(firstsel[0]) if firstsel and firstsel[0] is not None else len(first)
        _check(lib().cofactor_agg_update_host(
            self._h, _ptr_array([c.ctypes.data for c in num]), _ptr_array([c.ctypes.data for c in cat]),
            ns, cs, None if ri is None else ri.ctypes.data, rows))

    def update_triples(self, blobs):
        offs = np.zeros(len(blobs) + 1, dtype=np.uint64)
        for i, b in enumerate(blobs):
            offs[i + 1] = offs[i] + len(b)
        flat = np.ascontiguousarray(np.concatenate(blobs) if blobs else np.zeros(0), dtype=np.float64)
        _check(lib().cofactor_agg_update_triples(self._h, flat.ctypes.data, offs.ctypes.data, len(blobs)))

    def combine(self, other):
        _check(lib().cofactor_agg_combine(self._h, other._h))

    def finalize(self):
        return _two_call(lib().cofactor_agg_finalize, self._h)

    def dense_len(self):
        return lib().cofactor_dense_len(self.n, self.kind)

    def export_dense_device(self, ptr):
        _check(lib().cofactor_agg_export_dense_device(self._h, ptr))

    def import_dense_device(self, ptr):
        _check(lib().cofactor_agg_import_dense_device(self._h, ptr))

    # ---- dictionary-aligned table seam (multi-GPU) ----
    def keys(self):
        """-> (keys int32[total], offsets uint64[m+1]): the state's keys per column, ascending."""
        need = C.c_uint64(0)
        offs = np.zeros(self.m + 1, dtype=np.uint64)
        _check(lib().cofactor_agg_keys(self._h, None, 0, C.byref(need), None))
        out = np.empty(max(1, need.value), dtype=np.int32)
        _check(lib().cofactor_agg_keys(self._h, out.ctypes.data, out.size, C.byref(need), offs.ctypes.data))
        return out[:need.value], offs

    def dict_signature(self):
        sig = C.c_uint64(0)
        _check(lib().cofactor_agg_dict_signature(self._h, C.byref(sig)))
        return sig.value

    def align_keys(self, keys, offsets):
        k = np.ascontiguousarray(keys, dtype=np.int32)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        assert o.size == self.m + 1
        _check(lib().cofactor_agg_align_keys(self._h, k.ctypes.data if k.size else None, o.ctypes.data))

    def tables_len(self):
        return lib().cofactor_agg_tables_len(self._h)

    def export_tables_device(self, ptr):
        _check(lib().cofactor_agg_export_tables_device(self._h, ptr))

    def import_tables_device(self, ptr):
        _check(lib().cofactor_agg_import_tables_device(self._h, ptr))

    # ---- pair tables kept as sorted lists (very high cardinalities) across ranks ----
    def sparse_lens(self):
        """-> uint64[m(m+1)/2]: entries of every pair's sorted list (0 for dense pair tables)."""
        np_ = self.m * (self.m + 1) // 2
        out = np.zeros(max(1, np_), dtype=np.uint64)
        _check(lib().cofactor_agg_sparse_lens(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size))
        return out[:np_]

    def sparse_is_list(self, pair):
        v = C.c_int32(0)
        _check(lib().cofactor_agg_sparse_is_list(self._h, pair, C.byref(v)))
        return bool(v.value)

    def sparse_export_device(self, pair, keys_ptr, counts_ptr):
        _check(lib().cofactor_agg_sparse_export_device(self._h, pair, keys_ptr, counts_ptr))

    def sparse_assign_device(self, pair, keys_ptr, counts_ptr, length):
        _check(lib().cofactor_agg_sparse_assign_device(self._h, pair, keys_ptr, counts_ptr, length))

    def allreduce(self, comm):
        """The whole multi-GPU seam below the C ABI (cofactor_agg_allreduce): afterwards this state is
        the merge of all ranks' states."""
        _check(lib().cofactor_agg_allreduce(self._h, comm._h))


COMM_ID_BYTES = 128


def comm_unique_id():
    """128 bytes rank 0 hands to the other ranks (cofactor_comm_unique_id)."""
    buf = (C.c_ubyte * COMM_ID_BYTES)()
    _check(lib().cofactor_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """The library's own RCCL communicator of one rank (cofactor_comm)."""

    def __init__(self, ctx, uid, rank, world):
        assert len(uid) == COMM_ID_BYTES
        h = C.c_void_p()
        buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(uid)
        _check(lib().cofactor_comm_create(ctx._h, buf, rank, world, C.byref(h)))
        self._h, self.ctx, self.rank, self.world = h, ctx, rank, world

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            lib().cofactor_comm_destroy(self._h)
        self._h = None

    def __del__(self):
        if lib is not None:
            self.close()


def lift_host(num_cols, cat_cols, kind=TRIPLE):
    """to_cofactor / to_nb_agg: list of one blob per row."""
    num = [np.ascontiguousarray(c, dtype=np.float32) for c in num_cols]
    cat = [np.ascontiguousarray(c, dtype=np.int32) for c in cat_cols]
    rows = len(num[0]) if num else (len(cat[0]) if cat else 0)
    offs = np.zeros(rows + 1, dtype=np.uint64)
    np_, cp_ = _ptr_array([c.ctypes.data for c in num]), _ptr_array([c.ctypes.data for c in cat])
    need = C.c_uint64(0)
    _check(lib().cofactor_lift_host(np_, len(num), cp_, len(cat), rows, kind, None, 0, C.byref(need), None))
    out = np.empty(need.value, dtype=np.float64)
    _check(lib().cofactor_lift_host(np_, len(num), cp_, len(cat), rows, kind, out.ctypes.data, out.size,
                                    C.byref(need), offs.ctypes.data))
    return [out[int(offs[i]):int(offs[i + 1])].copy() for i in range(rows)]


def _binary(fn, a, b, bound=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    if bound is not None:                             # a buffer that surely holds the result: ONE call
        need = C.c_uint64(0)
        out = np.empty(bound, dtype=np.float64)
        st = fn(a.ctypes.data, a.size, b.ctypes.data, b.size, out.ctypes.data, out.size, C.byref(need))
        if st == OK:
            return out[:need.value].copy()
        if st != ERR_CAPACITY:
            _check(st)
    return _two_call(fn, a.ctypes.data, a.size, b.ctypes.data, b.size)


def multiply(a, b):
    return _binary(lib().cofactor_triple_multiply, a, b)


def add(a, b):
    return _binary(lib().cofactor_triple_add, a, b, bound=len(a) + len(b))


def sub(a, b):
    return _binary(lib().cofactor_triple_sub, a, b, bound=len(a) + len(b))


def _two_call_f32(fn, *args):
    """The trainers' two-call protocol WITHOUT training twice: a buffer that holds any usual parameter
    vector goes into the first call (a size query runs the whole training to learn the length); only
    a longer result costs the second call."""
    need = C.c_uint64(0)
    out = np.empty(1 << 16, dtype=np.float32)
    st = fn(*args, out.ctypes.data, out.size, C.byref(need))
    if st == ERR_CAPACITY:
        out = np.empty(need.value, dtype=np.float32)
        st = fn(*args, out.ctypes.data, out.size, C.byref(need))
    _check(st)
    return out[:need.value].copy()


def linreg_train(triple, label, step_size=0.001, lam=0.0, max_iterations=10000,
                 compute_variance=False, normalize=False):
    """linreg_train(triple, label, learning_rate, regularization, max_iterations,
    include_variance, normalize) -> float32 parameter vector (host; no GPU involved)."""
    b = np.ascontiguousarray(triple, dtype=np.float64)
    return _two_call_f32(lib().cofactor_linreg_train, b.ctypes.data, b.size, label, step_size, lam,
                         max_iterations, int(compute_variance), int(normalize))


def lda_train(triple, label, shrinkage=0.0, normalize=False):
    """lda_train(triple, label, shrinkage, normalize) -> float32 parameter vector (host)."""
    b = np.ascontiguousarray(triple, dtype=np.float64)
    return _two_call_f32(lib().cofactor_lda_train, b.ctypes.data, b.size, label, shrinkage, int(normalize))


def to_text(blob, aggregate_names=True):
    """The triple as DuckDB's STRUCT literal text (what the reference's MICE drivers paste into SQL)."""
    b = np.ascontiguousarray(blob, dtype=np.float64)
    need = C.c_uint64(0)
    _check(lib().cofactor_triple_to_text(b.ctypes.data, b.size, int(aggregate_names), None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value)
    _check(lib().cofactor_triple_to_text(b.ctypes.data, b.size, int(aggregate_names), buf, need.value, C.byref(need)))
    return buf.value.decode()


def from_text(text):
    raw = text.encode()
    return _two_call(lib().cofactor_triple_from_text, raw, len(raw))


def blob_len(blob):
    b = np.ascontiguousarray(blob, dtype=np.float64)
    return lib().cofactor_blob_len(b.ctypes.data, b.size)
