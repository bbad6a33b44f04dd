"""Multi-GPU seam of the aggregate: one process per GPU, rows sharded with no
halo, ONE all-reduce of the dense partial triple (RCCL over xGMI when the backend is "nccl"), and
the sparse categorical lists merged on the host from an all-gather of finalised blobs.

Works with any initialised torch.distributed backend: "nccl" (= RCCL) on GPUs, "gloo" for the
world-size-2 CPU tests (tests/test_dist_gloo.py), where the per-rank blobs come from elsewhere.
"""
import numpy as np

from . import add as _add_blobs


def shard_bounds(rows, rank, world):
    """Contiguous row range [lo, hi) of `rank` (shard g gets rows [g R/G, (g+1) R/G))."""
    return rows * rank // world, rows * (rank + 1) // world


def allreduce_dense(agg, dist, device):
    """In place: the aggregate's N / lin / quad become the totals over all ranks.
    GPU path: export to a device buffer -> dist.all_reduce(SUM) -> import."""
    import torch
    buf = torch.zeros(int(agg.dense_len()), dtype=torch.float64, device=device)
    agg.export_dense_device(buf.data_ptr())
    dist.all_reduce(buf)
    if buf.is_cuda:
        torch.cuda.synchronize(device)
    agg.import_dense_device(buf.data_ptr())


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
    total = np.ascontiguousarray(blobs[0], dtype=np.float64)
    for b in blobs[1:]:
        total = _add_blobs(total, b)
    return total


def allreduce_triple(agg, dist, device):
    """The reduced triple (flat blob) on every rank.  Dense part: one all-reduce on the device.
    Categorical part (m > 0): all-gather of the ranks' blobs, host merge in rank order; the
    dense fields of the merged blob are then replaced by the all-reduced ones so that every rank
    holds bit-identical values."""
    allreduce_dense(agg, dist, device)
    mine = agg.finalize()               # dense = global totals, lists = this rank's rows
    if agg.m == 0:
        return mine
    blobs = allgather_blobs(mine, dist, device=device)
    # each gathered blob carries the GLOBAL dense part; neutralise all but one before adding
    n = agg.n
    dense_len = int(agg.dense_len())
    for b in blobs[1:]:
        b[3:3 + dense_len] = 0.0
    merged = merge_blobs(blobs)
    merged[3:3 + dense_len] = mine[3:3 + dense_len]
    return merged
