#!/usr/bin/env python3
"""bench.py — rows/s of sum_to_triple_20_0 on MI355X (BASELINE.json metric), one process per GPU.

A "step" is one complete pass of the hot path over the rank's resident table: reset the
aggregate, cofactor_agg_update_device over all rows (HIP Gram kernel), for N > 1 one RCCL
all-reduce of the dense partial triple, and finalize to the host blob.  Inputs are synthetic
(uniform [0,1) float32 columns, seed 42) and already resident in HBM when the timed region starts.

    python bench.py                       # N = 1, 1e9 rows x 20 float columns (80 GB in HBM)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Weak scaling: every rank holds `--rows` rows (default 1e9), the table has N * rows rows.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "duckdb-imputation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=float, default=1e9, help="rows per GPU")
    ap.add_argument("--num-cols", type=int, default=20)
    ap.add_argument("--cat-cols", type=int, default=0)
    ap.add_argument("--keys", type=int, default=16, help="distinct keys per categorical column")
    ap.add_argument("--cpu-sample-rows", type=float, default=6e7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-dist", action="store_true", help="run the multi-GPU code path at the current world size")
    return ap.parse_args()


def make_table(torch, rows, n, m, keys, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    num = [torch.rand(rows, generator=g, device=device, dtype=torch.float32) for _ in range(n)]
    cat = [torch.randint(0, keys, (rows,), generator=g, device=device, dtype=torch.int32) for _ in range(m)]
    return num, cat


def cpu_baseline(torch, num, cat, sample_rows, n, m):
    """The oracle in faithful (float accumulator) mode — a port of the reference's update loop,
    std::map categoricals included — on a bounded prefix of the same table, thread-local states
    over contiguous shards merged by combine, as DuckDB runs the reference."""
    from oracle import oracle as orc
    rows = int(min(sample_rows, num[0].numel() if num else cat[0].numel()))
    h_num = [c[:rows].cpu().numpy() for c in num]
    h_cat = [c[:rows].cpu().numpy() for c in cat]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    st = orc.State(orc.FAITHFUL)
    t0 = time.perf_counter()
    st.update(h_num, h_cat, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": rows / dt, "unit": "rows/s", "cores": cores, "kind": "port",
            "sample": "first %d rows of the bench table, sum_to_triple_%d_%d, oracle faithful-fp32 "
                      "mode, %d threads (thread-local states + combine), %.2f s wall"
                      % (rows, n, m, cores, dt)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import cofactor_hip
    from cofactor_hip import dist as cdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run" % args.gpus)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # --rehearse-dist: take the N > 1 code path (RCCL process group, all-reduce of the partial
    # triple, barriers) with whatever world size there is, e.g. 1 on a one-GPU box
    use_dist = world > 1 or args.rehearse_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)

    rows, n, m = int(args.rows), args.num_cols, args.cat_cols
    num, cat = make_table(torch, rows, n, m, args.keys, device, seed=42 + rank)
    ctx = cofactor_hip.Context(local_rank)
    agg = ctx.aggregate(n, m)
    num_ptrs = [t.data_ptr() for t in num]
    cat_ptrs = [t.data_ptr() for t in cat]
    torch.cuda.synchronize()

    def step():
        agg.reset()
        agg.update_device_ptrs(num_ptrs, cat_ptrs, rows)
        if use_dist:
            # ONE RCCL all-reduce of the dense partial triple (+ host merge of categorical lists)
            return cdist.allreduce_triple(agg, dist, device)
        return agg.finalize()

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.profile(True)
    ctx.profile_read()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        blob = step()
    fence()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile(False)

    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    total_rows = rows * world
    assert int(blob[3]) == total_rows, (blob[3], total_rows)      # N of the reduced triple
    value = total_rows * args.steps / dt

    if rank == 0:
        bytes_per_row = 4 * (n + m)
        cands = [("gram_kernel", prof["gram_ms"], prof["gram_launches"], 4 * n * rows),
                 ("cat_accumulate_kernel", prof["cat_ms"], prof["cat_launches"], bytes_per_row * rows),
                 ("fused_kernel", prof["fused_ms"], prof["fused_launches"], bytes_per_row * rows)]
        kname, kms, kl, kbytes = max(cands, key=lambda c: c[1])       # the dominant kernel
        avg_ms = kms / max(1, kl)
        achieved = kbytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = "%s_%d_%d" % (kname, n, m)
                # the committed PMC figure is per launch at the row count it was measured with
                if tj.get(key + "__rows") == rows:
                    traffic = tj.get(key)
            except Exception:
                traffic = None
        out = {
            "metric": "rows/sec on sum_to_triple_%d_%d" % (n, m),
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "sum_to_triple_%d_%d over %d rows per GPU (%d total), uniform[0,1) "
                                   "float32 columns%s, resident in HBM" %
                                   (n, m, rows, total_rows, (", %d int32 columns with %d keys" % (m, args.keys)) if m else ""),
                       "rows_per_gpu": rows, "num_cols": n, "cat_cols": m,
                       "arithmetic": "f32 inputs and MFMA products, f64 accumulation, exact integer counts",
                       "parallelism": "row-sharded x%d, one RCCL all-reduce of the dense partial triple" % world},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_kernel_ms": avg_ms, "launches": kl,
                         "algorithmic_bytes_per_launch": kbytes},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(torch, num, cat, args.cpu_sample_rows, n, m)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    agg.close()
    ctx.close()


if __name__ == "__main__":
    main()
