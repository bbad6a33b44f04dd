"""world_size = 2, gloo, CPU: the N > 1 logic that does not need a GPU — row sharding, the
variable-length all-gather of finalised blobs and their host merge (cofactor_triple_add).
Per-rank blobs come from the oracle here (test infrastructure); on GPUs they come from the HIP
aggregate (bench.py, tests/test_gpu_parity.py::test_dense_export_import_roundtrip)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _table(rows, n, m, seed):
    rng = np.random.default_rng(seed)
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    cat = [rng.integers(-3, 4 + 3 * c, rows).astype(np.int32) for c in range(m)]
    return num, cat


def _worker(rank, world, port, rows, n, m, out_dir):
    for p in (ROOT, os.path.join(ROOT, "duckdb-imputation_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    from cofactor_hip import dist as cdist
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    num, cat = _table(rows, n, m, seed=5)
    lo, hi = cdist.shard_bounds(rows, rank, world)
    mine = orc.State(orc.WIDE).update([c[lo:hi] for c in num], [c[lo:hi] for c in cat]).finalize()
    blobs = cdist.allgather_blobs(mine, dist)
    merged = cdist.merge_blobs(blobs)
    np.save(os.path.join(out_dir, "merged_%d.npy" % rank), merged)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,m", [(3, 2), (20, 0), (0, 3)])
def test_two_rank_merge_equals_whole_table(tmp_path, n, m):
    from oracle import oracle as orc
    from triple_fmt import blob_to_dict
    rows, world = 10_001, 2
    mp.spawn(_worker, args=(world, _free_port(), rows, n, m, str(tmp_path)), nprocs=world, join=True)
    num, cat = _table(rows, n, m, seed=5)
    whole = blob_to_dict(orc.State(orc.WIDE).update(num, cat).finalize())
    for r in range(world):
        got = blob_to_dict(np.load(os.path.join(str(tmp_path), "merged_%d.npy" % r)))
        assert got == whole          # integer-valued table: exact in any merge order


def test_shard_bounds_cover_rows_without_overlap():
    from cofactor_hip import dist as cdist
    for rows in (0, 1, 7, 1000, 10**9 + 7):
        for world in (1, 2, 3, 8):
            b = [cdist.shard_bounds(rows, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == rows
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
