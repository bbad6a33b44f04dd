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


def _rccl_worker(rank, world, port, rows, n, m, keys, out_dir, use_lib_comm):
    for p in (ROOT, os.path.join(ROOT, "duckdb-imputation_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import cofactor_hip
    from cofactor_hip import dist as cdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    device = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    num, cat = _table(rows, n, m, 5, keys)
    lo, hi = cdist.shard_bounds(rows, rank, world)
    d_num = [torch.from_numpy(c[lo:hi]).to(device) for c in num]
    d_cat = [torch.from_numpy(c[lo:hi]).to(device) for c in cat]
    torch.cuda.synchronize()
    ctx = cofactor_hip.Context(rank)
    comm = cdist.make_comm(ctx, dist) if use_lib_comm else None
    agg = ctx.aggregate(n, m)
    agg.update_device(d_num, d_cat)
    np.save(os.path.join(out_dir, "merged_%d.npy" % rank), cdist.allreduce_triple(agg, dist, device, comm))
    agg.reset()
    agg.update_device(d_num, d_cat)
    np.save(os.path.join(out_dir, "again_%d.npy" % rank), cdist.allreduce_triple(agg, dist, device, comm))
    dist.barrier()
    if comm is not None:
        comm.close()
    agg.close()
    ctx.close()
    dist.destroy_process_group()


def _two_gpus():
    import torch
    return torch.cuda.device_count() >= 2


@pytest.mark.parametrize("use_lib_comm", [True, False])
@pytest.mark.parametrize("n,m,keys,rows", [(20, 0, 0, 300_001), (10, 10, 16, 300_001), (4, 3, 40, 300_001)])
def test_rccl_all_reduce_on_two_gpus(tmp_path, n, m, keys, rows, use_lib_comm):
    """The real collective: two ranks on two GPUs, RCCL over xGMI — through the library's own
    communicator (cofactor_agg_allreduce) and through torch.distributed's nccl backend on the
    library's stream.  Skipped on a one-GPU box; the assertion is the gloo test's: every rank ends up
    with the whole table's triple."""
    if not _two_gpus():
        pytest.skip("needs two GPUs")
    from oracle import oracle as orc
    from triple_fmt import blob_to_dict
    world = 2
    mp.spawn(_rccl_worker, args=(world, _free_port(), rows, n, m, keys, str(tmp_path), use_lib_comm), nprocs=world, join=True)
    num, cat = _table(rows, n, m, 5, keys)
    whole = blob_to_dict(orc.State(orc.WIDE).update(num, cat).finalize())
    for r in range(world):
        for name in ("merged", "again"):
            assert blob_to_dict(np.load(os.path.join(str(tmp_path), "%s_%d.npy" % (name, r)))) == whole, (name, r)


def _rccl_big_worker(rank, world, port, rows_per_rank, out_dir):
    for p in (ROOT, os.path.join(ROOT, "duckdb-imputation_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import cofactor_hip
    from cofactor_hip import dist as cdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    device = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    g = torch.Generator(device=device).manual_seed(77 + rank)
    ints = [torch.randint(0, 8, (rows_per_rank,), generator=g, device=device, dtype=torch.int32) for _ in range(20)]
    cols = [x.float() for x in ints]
    torch.cuda.synchronize()
    ctx = cofactor_hip.Context(rank)
    comm = cdist.make_comm(ctx, dist)
    agg = ctx.aggregate(20, 0)
    agg.update_device(cols, [])
    blob = cdist.allreduce_triple(agg, dist, device, comm)
    lin = torch.stack([x.sum(dtype=torch.int64) for x in ints])
    q01 = (ints[0] * ints[1]).sum(dtype=torch.int64).reshape(1)
    ref = torch.cat([lin, q01])
    dist.all_reduce(ref)
    np.save(os.path.join(out_dir, "blob_%d.npy" % rank), blob)
    np.save(os.path.join(out_dir, "ref_%d.npy" % rank), ref.cpu().numpy())
    dist.barrier()
    comm.close(); agg.close(); ctx.close()
    dist.destroy_process_group()


def test_rccl_20_0_at_1e8_rows_per_rank_on_two_gpus(tmp_path):
    """BASELINE's configs[3] in small: sum_to_triple_20_0, 1e8 integer-valued rows per rank, two GPUs,
    the library's communicator: N, every lin_agg entry and quad[0,1] equal the int64 sums over both
    ranks exactly."""
    if not _two_gpus():
        pytest.skip("needs two GPUs")
    world, rows = 2, 100_000_000
    mp.spawn(_rccl_big_worker, args=(world, _free_port(), rows, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        blob = np.load(os.path.join(str(tmp_path), "blob_%d.npy" % r))
        ref = np.load(os.path.join(str(tmp_path), "ref_%d.npy" % r))
        assert blob[3] == world * rows
        assert np.array_equal(blob[4:24], ref[:20].astype(np.float64))
        assert blob[24 + 1] == float(ref[20])                         # quad row 0: (0,0), (0,1), ...


def _sparse_worker(rank, world, port, rows, out_dir):
    for p in (ROOT, os.path.join(ROOT, "duckdb-imputation_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["COFACTOR_SPARSE_CELLS"] = "2000"
    import torch
    import torch.distributed as dist
    import cofactor_hip
    from cofactor_hip import dist as cdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    device = torch.device("cuda", 0)
    num, cat = _sparse_table(rows)
    lo, hi = cdist.shard_bounds(rows, rank, world)
    d_num = [torch.from_numpy(c[lo:hi]).to(device) for c in num]
    d_cat = [torch.from_numpy(c[lo:hi]).to(device) for c in cat]
    torch.cuda.synchronize()
    ctx = cofactor_hip.Context(0)
    agg = ctx.aggregate(len(num), len(cat))
    agg.update_device(d_num, d_cat)
    np.save(os.path.join(out_dir, "merged_%d.npy" % rank), cdist.allreduce_triple(agg, dist, device))
    np.save(os.path.join(out_dir, "lens_%d.npy" % rank), agg.sparse_lens())
    dist.barrier()
    agg.close(); ctx.close()
    dist.destroy_process_group()


def _sparse_table(rows):
    rng = np.random.default_rng(8)
    num = [rng.integers(0, 9, rows).astype(np.float32) for _ in range(2)]
    cat = [rng.integers(-40, 160, rows).astype(np.int32), rng.integers(0, 12, rows).astype(np.int32),
           rng.integers(0, 300, rows).astype(np.int32)]
    return num, cat


def test_two_ranks_with_pair_tables_kept_as_sorted_lists(tmp_path):
    """States whose big pair tables are sorted lists take part in the seam: dense tables are aligned
    and all-reduced, the lists gathered and merged.  Threshold lowered so that
    200 x 200, 200 x 300 and 300 x 300 keys are lists and the pairs with the 12-key column dense."""
    from oracle import oracle as orc
    from triple_fmt import blob_to_dict
    rows, world = 200_001, 2
    mp.spawn(_sparse_worker, args=(world, _free_port(), rows, str(tmp_path)), nprocs=world, join=True)
    num, cat = _sparse_table(rows)
    whole = blob_to_dict(orc.State(orc.WIDE).update(num, cat).finalize())
    for r in range(world):
        assert blob_to_dict(np.load(os.path.join(str(tmp_path), "merged_%d.npy" % r))) == whole
        assert int(np.load(os.path.join(str(tmp_path), "lens_%d.npy" % r)).sum()) > 0


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
                          {"x0": d(x0n.astype(np.uint8))}, {"k0": d(k0n.astype(np.uint8))}, first_row=lo)


def _mice_worker(rank, world, port, rows, out_dir, partitioned=False):
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
    if partitioned:
        models, part = mice.run_mice_partitioned(ctx, t, iterations=1, seed=3, dist=dist, device=device)
        part.write_back()
    else:
        models = mice.run_mice(ctx, t, iterations=1, seed=3, dist=dist, device=device)
    np.save(os.path.join(out_dir, "k0_%d.npy" % rank), models["k0"])
    np.save(os.path.join(out_dir, "x0_%d.npy" % rank), models["x0"])
    np.save(os.path.join(out_dir, "k0col_%d.npy" % rank), t.cat["k0"].cpu().numpy())
    np.save(os.path.join(out_dir, "x0col_%d.npy" % rank), t.num["x0"].cpu().numpy())
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("partitioned", [False, True])
def test_two_rank_mice_trains_the_models_of_the_single_process_run(tmp_path, partitioned):
    """The MICE loop over two row shards: every rank aggregates its shard, the all-reduced triple is
    the whole table's, so both ranks train the single-process run's models and fill in the same
    keys (the key column's fill is deterministic)."""
    import torch
    import cofactor_hip
    from cofactor_hip import mice
    rows, world = 120_000, 2
    mp.spawn(_mice_worker, args=(world, _free_port(), rows, str(tmp_path), partitioned), nprocs=world, join=True)
    t = _mice_table(rows, 0, rows, torch.device("cuda", 0))
    ctx = cofactor_hip.Context(0)
    models = mice.run_mice(ctx, t, iterations=1, seed=3)
    k0 = t.cat["k0"].cpu().numpy()
    x0 = t.num["x0"].cpu().numpy()
    ctx.close()
    filled = []
    for r in range(world):
        for name in ("k0", "x0"):
            got = np.load(os.path.join(str(tmp_path), "%s_%d.npy" % (name, r)))
            assert np.allclose(got, models[name], rtol=1e-4, atol=1e-5), (name, r)
        filled.append(np.load(os.path.join(str(tmp_path), "k0col_%d.npy" % r)))
    if partitioned:                                  # (an argmax within the rounding of a blob subtraction may flip)
        assert np.mean(np.concatenate(filled) == k0) > 0.999
    else:
        assert np.array_equal(np.concatenate(filled), k0)
    # the numeric column too: the imputation noise of a row depends on its place in the whole table,
    # not on the sharding (the trained parameters agree to 1e-4, so do the filled values)
    x0_sharded = np.concatenate([np.load(os.path.join(str(tmp_path), "x0col_%d.npy" % r)) for r in range(world)])
    same = np.concatenate(filled) == k0
    assert np.allclose(x0_sharded[same], x0[same], rtol=2e-3, atol=2e-3)
    assert np.abs(x0_sharded[same] - x0[same]).max() < 0.05
