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


def _mice_table(rows, lo, hi, device):
    import torch
    rng = np.random.default_rng(3)
    x1 = rng.normal(size=rows).astype(np.float32)
    x2 = rng.normal(size=rows).astype(np.float32)
    k1 = (rng.integers(0, 4, rows) * 3 + 5).astype(np.int32)
    k0 = ((((x1 + 0.5 * x2 + 0.3 * rng.normal(size=rows)) > 0).astype(np.int32) + (k1 > 8) * 2) * 10 + 1).astype(np.int32)
    x0 = (2.0 * x1 - x2 + 0.7 * (k1 == 8) + 0.1 * rng.normal(size=rows)).astype(np.float32)
    x0n, k0n = rng.random(rows) < 0.1, rng.random(rows) < 0.1
    from cofactor_hip import mice
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a[lo:hi])).to(device)
    return mice.MiceTable({"x0": d(np.where(x0n, np.float32(-999), x0)), "x1": d(x1), "x2": d(x2)},
                          {"k0": d(np.where(k0n, np.int32(-999), k0)), "k1": d(k1)},
                          {"x0": d(x0n.astype(np.uint8))}, {"k0": d(k0n.astype(np.uint8))})


def _mice_worker(rank, world, port, rows, out_dir):
    for p in (ROOT, os.path.join(ROOT, "duckdb-imputation_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import cofactor_hip
    from cofactor_hip import dist as cdist
    from cofactor_hip import mice
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    device = torch.device("cuda", 0)
    lo, hi = cdist.shard_bounds(rows, rank, world)
    t = _mice_table(rows, lo, hi, device)
    ctx = cofactor_hip.Context(0)
    models = mice.run_mice(ctx, t, iterations=1, seed=3, dist=dist, device=device)
    np.save(os.path.join(out_dir, "k0_%d.npy" % rank), models["k0"])
    np.save(os.path.join(out_dir, "x0_%d.npy" % rank), models["x0"])
    np.save(os.path.join(out_dir, "k0col_%d.npy" % rank), t.cat["k0"].cpu().numpy())
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


def test_two_rank_mice_trains_the_models_of_the_single_process_run(tmp_path):
    """The MICE loop over two row shards: every rank aggregates its shard, the all-reduced triple is
    the whole table's, so both ranks train the single-process run's models and fill in the same
    keys (the key column's fill is deterministic)."""
    import torch
    import cofactor_hip
    from cofactor_hip import mice
    rows, world = 120_000, 2
    mp.spawn(_mice_worker, args=(world, _free_port(), rows, str(tmp_path)), nprocs=world, join=True)
    t = _mice_table(rows, 0, rows, torch.device("cuda", 0))
    ctx = cofactor_hip.Context(0)
    models = mice.run_mice(ctx, t, iterations=1, seed=3)
    k0 = t.cat["k0"].cpu().numpy()
    ctx.close()
    filled = []
    for r in range(world):
        for name in ("k0", "x0"):
            got = np.load(os.path.join(str(tmp_path), "%s_%d.npy" % (name, r)))
            assert np.allclose(got, models[name], rtol=1e-4, atol=1e-5), (name, r)
        filled.append(np.load(os.path.join(str(tmp_path), "k0col_%d.npy" % r)))
    assert np.array_equal(np.concatenate(filled), k0)
