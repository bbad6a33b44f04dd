"""The N > 1 path on real aggregates: two ranks (two processes, each with its own context on GPU
0 — a one-GPU box has no second card) shard one table, run the HIP aggregate on their rows and
merge through cofactor_hip.dist.allreduce_state: dictionary alignment, export kernel, ONE
all-reduce, import kernel.  Two processes cannot share a GPU under RCCL, so the collective itself
runs on gloo with the exported buffer staged through host memory; everything else is the code
the 8-GPU run executes.  The result on every rank must equal the whole-table oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _table(rows, n, m, seed, keys):
    rng = np.random.default_rng(seed)
    num = [rng.integers(0, 16, rows).astype(np.float32) for _ in range(n)]
    # rank-disjoint key ranges in the first column: the ranks' dictionaries differ
    cat = [rng.integers(-3, keys + 3 * c, rows).astype(np.int32) for c in range(m)]
    if m:
        cat[0][: rows // 2] = rng.integers(100, 100 + keys, rows // 2)
    return num, cat


def _worker(rank, world, port, rows, n, m, keys, nb, out_dir):
    for p in (ROOT, os.path.join(ROOT, "duckdb-imputation_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import cofactor_hip
    from cofactor_hip import dist as cdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    device = torch.device("cuda", 0)
    num, cat = _table(rows, n, m, 5, keys)
    lo, hi = cdist.shard_bounds(rows, rank, world)
    d_num = [torch.from_numpy(c[lo:hi]).to(device) for c in num]
    d_cat = [torch.from_numpy(c[lo:hi]).to(device) for c in cat]
    torch.cuda.synchronize()
    ctx = cofactor_hip.Context(0)
    agg = ctx.aggregate(n, m, cofactor_hip.NB if nb else cofactor_hip.TRIPLE)
    agg.update_device(d_num, d_cat)
    first = cdist.allreduce_triple(agg, dist, device)
    np.save(os.path.join(out_dir, "merged_%d.npy" % rank), first)
    # steady state: same dictionaries -> no key exchange, same result after a reset + re-scan
    sig = agg.dict_signature()
    agg.reset()
    agg.update_device(d_num, d_cat)
    exchanged = cdist.align_dictionaries(agg, dist, "cpu")
    again = cdist.allreduce_triple(agg, dist, device)
    np.save(os.path.join(out_dir, "again_%d.npy" % rank), again)
    np.save(os.path.join(out_dir, "flags_%d.npy" % rank), np.array([int(exchanged), int(sig != 0 or m == 0)]))
    dist.barrier()
    agg.close()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,m,keys,nb", [(20, 0, 0, False), (3, 2, 5, False), (10, 10, 16, False),
                                          (4, 3, 40, False), (0, 3, 7, False), (5, 2, 9, True)])
def test_two_ranks_allreduce_equals_whole_table(tmp_path, n, m, keys, nb):
    from oracle import oracle as orc
    from triple_fmt import blob_to_dict
    rows, world = 300_001, 2
    mp.spawn(_worker, args=(world, _free_port(), rows, n, m, keys, nb, str(tmp_path)), nprocs=world, join=True)
    num, cat = _table(rows, n, m, 5, keys)
    whole = blob_to_dict(orc.State(orc.WIDE).update(num, cat, nb=nb).finalize())
    for r in range(world):
        for name in ("merged", "again"):
            got = blob_to_dict(np.load(os.path.join(str(tmp_path), "%s_%d.npy" % (name, r))))
            assert got == whole, name          # integer-valued table: exact in any merge order
        exchanged, aligned = np.load(os.path.join(str(tmp_path), "flags_%d.npy" % r))
        assert aligned == 1 and exchanged == 0   # the second round needed no key exchange
