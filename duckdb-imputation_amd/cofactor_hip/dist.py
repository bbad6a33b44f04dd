"""Multi-GPU seam of the aggregate: one process per GPU, rows sharded with no
halo, and the ranks' partial triples merged by ONE all-reduce (RCCL over xGMI when the backend is
"nccl") of one device buffer

    [ N, lin, quad | cnt | s | p ]

— the dense part (Triple::SumStateCombine, sum_state.cpp:25,73-83) followed by the categorical
tables re-indexed to a dictionary all ranks share (sum_state.cpp:87-111 as a dense sum).  Nothing
travels through the host: export kernel -> all-reduce -> import kernel are chained on the
library's own stream, which torch sees as an ExternalStream.

Before the all-reduce the ranks make sure their dictionaries agree: a tiny all-gather of
(signature, #keys per column); only if some rank met a new key since the last alignment are the key
lists gathered and cofactor_agg_align_keys run.  In the steady state (a MICE loop over a fixed
table) a step is one small all-gather and one all-reduce.

Backends: "nccl" (= RCCL) on GPUs.  "gloo" stages the buffer through host memory — that is how the
world-size-2 tests run two ranks on one GPU.

The same seam exists BELOW the C ABI (cofactor_comm_* / cofactor_agg_allreduce, csrc/comm.cpp: the
library's own RCCL communicator, no torch in the data path): `make_comm` + `allreduce_state(...,
comm=comm)`.  torch.distributed is then only the rendezvous that ships the 128-byte id.
"""
import numpy as np


def shard_bounds(rows, rank, world):
    """Contiguous row range [lo, hi) of `rank` (shard g gets rows [g R/G, (g+1) R/G))."""
    return rows * rank // world, rows * (rank + 1) // world


def _seam_buffer(agg, length, device):
    """Persistent device buffer of the aggregate (never handed back to torch's allocator while
    kernels of the library's stream may still read it)."""
    import torch
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:        # 'cuda' never equals 'cuda:0': the buffer
        device = torch.device("cuda", torch.cuda.current_device())   # would be re-made on every call
    buf = getattr(agg, "_seam_buf", None)
    if buf is None or buf.numel() < length or buf.device != device:
        agg.ctx.synchronize()                       # (the import kernel of the last step may still read the old one)
        buf = torch.empty(max(256, int(length)), dtype=torch.float64, device=device)
        agg._seam_buf = buf
    return buf[:length]


def _ctx_stream(agg, device):
    import torch
    st = getattr(agg.ctx, "_torch_stream", None)
    if st is None:
        st = torch.cuda.ExternalStream(agg.ctx.stream, device=device)
        agg.ctx._torch_stream = st
    return st


def _gather_int64(vec, dist, device):
    """all_gather of equally sized int64 vectors -> list of numpy arrays, one per rank."""
    import torch
    mine = torch.as_tensor(np.ascontiguousarray(vec, dtype=np.int64), device=device)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [o.cpu().numpy() for o in out]


def align_dictionaries(agg, dist, comm_device):
    """SURVEY.md §8e steps 1-2.  Returns True if the key lists had to be exchanged."""
    if agg.m == 0:
        return False
    sig = agg.dict_signature()
    keys = offs = None
    if sig == 0:
        keys, offs = agg.keys()
        sizes = np.diff(offs.astype(np.int64))
    else:
        sizes = np.zeros(agg.m, dtype=np.int64)
    # signature as two 31-bit halves + a third word: int64 all_gather keeps all 64 bits, split only
    # to stay clear of sign handling
    head = np.array([sig & 0x7FFFFFFF, (sig >> 31) & 0x7FFFFFFF, sig >> 62], dtype=np.int64)
    got = _gather_int64(np.concatenate([head, sizes]), dist, comm_device)
    sigs = {tuple(g[:3]) for g in got}
    if len(sigs) == 1 and sig != 0:
        return False                                   # every rank still holds the common dictionary
    if keys is None:
        keys, offs = agg.keys()
    # (ranks that were aligned reported zero sizes: gather the real ones)
    sizes_all = _gather_int64(np.diff(offs.astype(np.int64)), dist, comm_device)
    longest = int(max(int(s.sum()) for s in sizes_all))
    padded = np.zeros(max(1, longest), dtype=np.int64)
    padded[:keys.size] = keys
    lists = _gather_int64(padded, dist, comm_device)
    cols = [[] for _ in range(agg.m)]
    for lst, sz in zip(lists, sizes_all):
        pos = 0
        for c in range(agg.m):
            cols[c].append(lst[pos:pos + int(sz[c])])
            pos += int(sz[c])
    allk = [np.concatenate(c) if c else np.zeros(0, dtype=np.int64) for c in cols]
    goffs = np.zeros(agg.m + 1, dtype=np.uint64)
    goffs[1:] = np.cumsum([a.size for a in allk])
    agg.align_keys(np.concatenate(allk).astype(np.int32) if allk else np.zeros(0, np.int32), goffs)
    return True


def make_comm(ctx, dist):
    """The library's own communicator for this rank: rank 0 makes the id, torch.distributed (any
    backend) only carries its 128 bytes to the other ranks."""
    import cofactor_hip
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [cofactor_hip.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return cofactor_hip.Comm(ctx, box[0], rank, world)


def _gather_sparse_lists(agg, dist, device, comm_device):
    """Pair tables kept as sorted lists: every rank gathers all ranks' lists and merges them
    (cofactor_agg_sparse_*; the torch / gloo flavour of step 4 of cofactor_agg_allreduce)."""
    import torch
    if agg.m == 0 or agg.kind != 0:
        return
    lens = agg.sparse_lens().astype(np.int64)
    all_lens = _gather_int64(lens, dist, comm_device)
    for q in range(lens.size):
        total = int(sum(int(l[q]) for l in all_lens))
        if total == 0:
            continue
        longest = int(max(int(l[q]) for l in all_lens))
        mine = torch.zeros(2 * longest, dtype=torch.int64, device=device)
        torch.cuda.current_stream(mine.device).synchronize()
        agg.sparse_export_device(q, mine.data_ptr(), mine.data_ptr() + 8 * longest)
        agg.ctx.synchronize()
        send = mine.to(comm_device)
        got = [torch.empty_like(send) for _ in range(dist.get_world_size())]
        dist.all_gather(got, send)
        keys = torch.cat([g[:int(l[q])] for g, l in zip(got, all_lens)]).to(device)
        cnts = torch.cat([g[longest:longest + int(l[q])] for g, l in zip(got, all_lens)]).to(device)
        torch.cuda.current_stream(keys.device).synchronize()
        agg.sparse_assign_device(q, keys.data_ptr(), cnts.data_ptr(), total)


def allreduce_state(agg, dist, device, comm=None):
    """In place: the aggregate becomes the merge of all ranks' aggregates (dense totals and, after
    dictionary alignment, every categorical table).  One all-reduce.  With `comm` (make_comm) the
    whole seam runs below the C ABI on the library's own RCCL communicator."""
    import torch
    if comm is not None:
        agg.allreduce(comm)
        return
    backend = dist.get_backend()
    on_gpu = torch.device(device).type == "cuda"
    comm_device = device if (backend == "nccl" and on_gpu) else "cpu"
    # a rank that fails while preparing must not leave the others inside the collective: the ranks
    # exchange a status word (and the buffer length) first and raise together
    err = None
    try:
        align_dictionaries(agg, dist, comm_device)
        dlen, tlen = int(agg.dense_len()), int(agg.tables_len())
    except Exception as e:                          # noqa: BLE001 - reported on every rank below
        err, dlen, tlen = e, 0, 0
    got = _gather_int64(np.array([0 if err is None else 1, dlen + tlen], dtype=np.int64), dist, comm_device)
    bad = [r for r, g in enumerate(got) if int(g[0]) != 0]
    if bad:
        raise RuntimeError("allreduce_state: rank(s) %s could not prepare their state%s" %
                           (bad, "" if err is None else ": %s" % err))
    if len({int(g[1]) for g in got}) != 1:
        raise RuntimeError("allreduce_state: the ranks' aligned images differ in length: %s" % [int(g[1]) for g in got])
    buf = _seam_buffer(agg, dlen + tlen, device)
    dptr = buf.data_ptr()
    if backend == "nccl":
        with torch.cuda.stream(_ctx_stream(agg, device)):
            agg.export_dense_device(dptr)
            if tlen:
                agg.export_tables_device(dptr + 8 * dlen)
            dist.all_reduce(buf)                       # RCCL, ordered after / before our kernels by stream events
            agg.import_dense_device(dptr)
            if tlen:
                agg.import_tables_device(dptr + 8 * dlen)
    else:                                              # gloo: the same seam, staged through host memory
        agg.export_dense_device(dptr)
        if tlen:
            agg.export_tables_device(dptr + 8 * dlen)
        agg.ctx.synchronize()
        host = buf.cpu()
        dist.all_reduce(host)
        buf.copy_(host)
        torch.cuda.current_stream(buf.device).synchronize()
        agg.import_dense_device(dptr)
        if tlen:
            agg.import_tables_device(dptr + 8 * dlen)
    _gather_sparse_lists(agg, dist, device, comm_device)


def allreduce_dense(agg, dist, device):
    """Dense part only (m = 0 aggregates, or callers that merge the lists themselves)."""
    assert agg.m == 0, "allreduce_dense is for aggregates without key columns; use allreduce_state"
    allreduce_state(agg, dist, device)


def allreduce_triple(agg, dist, device, comm=None):
    """The reduced triple (flat blob), identical on every rank."""
    allreduce_state(agg, dist, device, comm)
    return agg.finalize()


# ---- blob-level helpers (CPU tests of the host merge; not on the GPU path any more) ----------------

def allgather_blobs(blob, dist, device="cpu"):
    """Every rank's finalised blob on every rank (variable length: sizes first, then padded)."""
    import torch
    world = dist.get_world_size()
    mine = torch.as_tensor(np.ascontiguousarray(blob, dtype=np.float64), device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([mine.numel()], dtype=torch.int64, device=device))
    longest = int(max(int(s.item()) for s in sizes))
    padded = torch.zeros(longest, dtype=torch.float64, device=device)
    padded[:mine.numel()] = mine
    out = [torch.zeros(longest, dtype=torch.float64, device=device) for _ in range(world)]
    dist.all_gather(out, padded)
    return [o[:int(s.item())].cpu().numpy() for o, s in zip(out, sizes)]


def merge_blobs(blobs):
    """Fold finalised triples in rank order with the library's Value-level add
    (cofactor_triple_add = Triple::sum_triple, imputation/triple/sum.cpp:68-209)."""
    from . import add as _add_blobs
    total = np.ascontiguousarray(blobs[0], dtype=np.float64)
    for b in blobs[1:]:
        total = _add_blobs(total, b)
    return total
