#!/usr/bin/env python3
"""bench.py — rows/s of sum_to_triple_20_0 over 1e9 rows on MI355X (BASELINE.json's metric), one
process per GPU.

A "step" is one complete pass of the hot path over the resident table: reset the aggregate,
cofactor_agg_update_device over the rank's rows (HIP kernels), for N > 1 ONE RCCL all-reduce of the
partial triple (export kernel -> all-reduce -> import kernel on the library's stream), and
finalize to the host blob.  Inputs are synthetic (cofactor_hip/synth.py: counter-based generator,
seed 42, uniform [0,1) float32 columns, so that every sharding is the SAME table) and already
resident in HBM when the timed region starts.

    python bench.py                       # N = 1: 1e9 rows x 20 float columns (80 GB in HBM)
    python bench.py --gpus N              # N > 1 from a plain shell: starts its own N ranks (below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process before this
process has imported torch or made any HIP call (a process that has touched the GPU must never be
replaced or forked), relays the ranks' output — rank 0's one JSON line — and exits with the
child's return code.

N > 1 is BASELINE.json's configs[3] — the SAME 1e9-row table sharded over the ranks ("scaling":
"strong"); `--scaling weak` gives every rank `--rows` rows instead.  After the timed loop the
reduced triple is checked against torch fp64 reductions of the same columns (all column sums and
a sample of the products, summed over the ranks) and its checksum is printed, so the runs at
N = 1, 2, 4, 8 can be compared value for value.  Rank 0 prints ONE JSON line.
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
SEED = 42


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--total-rows", type=float, default=1e9, help="rows of the whole table (strong scaling)")
    ap.add_argument("--rows", type=float, default=None, help="rows per GPU (implies --scaling weak)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default=None)
    ap.add_argument("--num-cols", type=int, default=20)
    ap.add_argument("--cat-cols", type=int, default=0)
    ap.add_argument("--keys", type=int, default=16, help="distinct keys per categorical column")
    ap.add_argument("--nb", action="store_true", help="sum_to_nb_agg instead of sum_to_triple")
    ap.add_argument("--cpu-sample-rows", type=float, default=6e7)
    ap.add_argument("--cpu-1t-sample-rows", type=float, default=2e7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-calibration", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--rehearse-dist", action="store_true", help="run the multi-GPU code path at the current world size")
    ap.add_argument("--collective", choices=["lib", "torch"], default="lib",
                    help="N > 1: the library's own RCCL communicator below the C ABI (cofactor_agg_allreduce) or "
                         "torch.distributed's all_reduce on the exported buffer")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="functional rehearsal of N > 1 on a one-GPU box: every rank on cuda:0, gloo instead of "
                         "RCCL (which refuses two ranks on one device); numbers mean nothing")
    return ap.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(num, cat, rows_mt, rows_1t, n, m, nb):
    """The oracle in faithful (float accumulator) mode — a port of the reference's update loop,
    std::map categoricals included (kind "port": the reference itself needs DuckDB headers that are
    not in the image) — on a bounded prefix of the same table: (i) one thread, (ii) one thread per
    host core with thread-local states over contiguous shards merged by combine, as DuckDB runs
    the reference."""
    from oracle import oracle as orc
    have = num[0].numel() if num else cat[0].numel()
    rows_mt, rows_1t = int(min(rows_mt, have)), int(min(rows_1t, have))
    h_num = [c[:rows_mt].cpu().numpy() for c in num]
    h_cat = [c[:rows_mt].cpu().numpy() for c in cat]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, cores)
    # per_row_pointers: every 2048-row chunk through a vector of state pointers read per row, as the
    # executor drives the reference's update (sum_no_lift.cpp:84,94,139) — SURVEY.md §8(d)'s baseline
    t0 = time.perf_counter()
    orc.State(orc.FAITHFUL).update([c[:rows_1t] for c in h_num], [c[:rows_1t] for c in h_cat], nb=nb,
                                   threads=1, per_row_pointers=True)
    dt1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.State(orc.FAITHFUL).update(h_num, h_cat, nb=nb, threads=cores, per_row_pointers=True)
    dtm = time.perf_counter() - t0
    name = "sum_to_%s_%d_%d" % ("nb_agg" if nb else "triple", n, m)
    return {"value": rows_mt / dtm, "unit": "rows/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(),
            "one_thread": {"value": rows_1t / dt1, "unit": "rows/s", "cores": 1,
                           "sample": "first %d rows, %.2f s wall" % (rows_1t, dt1)},
            "path": "per-row state pointers (one 2048-entry pointer vector per chunk, read per row)",
            "sample": "first %d rows of the bench table, %s, oracle (CPU restatement of the reference's "
                      "update loop) in faithful-fp32 mode, per-row state pointers, %d threads (thread-local "
                      "states + combine), %.2f s wall" % (rows_mt, name, cores, dtm)}


def calibrate(ctx):
    """What a plain streaming kernel reaches on this GPU (cofactor_ctx_calibrate: float4
    non-temporal copy and read-only stream over 4 GiB, the access shape of the Gram kernel)."""
    copy, read = ctx.calibrate(4 << 30, 5)
    return {"copy": copy, "read": read}


def reference_sums(torch, num, pairs):
    """fp64 column sums and the listed products of this rank's rows, by torch (chunked so the fp64
    copies stay small)."""
    n = len(num)
    lin = torch.zeros(n, dtype=torch.float64, device=num[0].device)
    quad = torch.zeros(len(pairs), dtype=torch.float64, device=num[0].device)
    rows = num[0].numel()
    step = 1 << 26
    for a in range(0, rows, step):
        d = [c[a:a + step].double() for c in num]
        for k in range(n):
            lin[k] += d[k].sum()
        for i, (j, k) in enumerate(pairs):
            quad[i] += torch.dot(d[j], d[k])
    return lin, quad


def self_launch(args, argv):
    """`python bench.py --gpus N` from a plain shell: start one rank per GPU with
    torch.distributed.run as a child process and relay its output and return code.  Nothing in
    this process has touched the GPU (torch is not even imported here)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    launcher = os.environ.get("BENCH_LAUNCH_MODULE", "torch.distributed.run")   # (tests put a stand-in here)
    cmd = [sys.executable, "-m", launcher, "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it here
    env.setdefault("OMP_NUM_THREADS", "8")
    if os.environ.get("BENCH_LAUNCH_DRY_RUN"):             # tests: show the command, start nothing
        print(json.dumps({"launch": cmd}))
        return 0
    # The merge of the ranks' triples runs on the library's own RCCL communicator by default.  Should
    # that attempt fail or stall (it has its own bootstrap: a second RCCL instance next to torch's),
    # the ranks are stopped and started once more on torch.distributed's all_reduce, so that a scaling
    # run still yields its line; `config.collective` says which collective was timed.
    explicit = any(a == "--collective" or a.startswith("--collective=") for a in argv)
    limit = float(os.environ.get("BENCH_LIB_TIMEOUT", "600"))
    if explicit:
        return subprocess.call(cmd, env=env)
    import signal

    def own_session():
        # a session of its own (so that a stalled attempt can be stopped as a group) that still ends
        # with this process: SIGTERM to the launcher when the parent dies, however it dies
        os.setsid()
        try:
            import ctypes
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGTERM)     # PR_SET_PDEATHSIG
        except Exception:                                  # noqa: BLE001
            pass

    proc = subprocess.Popen(cmd, env=env, preexec_fn=own_session)
    live = [proc]

    def relay(signum, _frame):                             # the driver stops us: stop the ranks too
        try:
            os.killpg(live[0].pid, signal.SIGTERM)
        except Exception:                                  # noqa: BLE001
            pass
        sys.exit(128 + signum)

    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, relay)
    try:
        rc = proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        rc = None
    if rc == 0:
        return 0
    if rc is None:                                         # our own process group, by its id
        try:
            os.killpg(proc.pid, signal.SIGTERM)
            proc.wait(timeout=30)
        except Exception:                                  # noqa: BLE001
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except Exception:                              # noqa: BLE001
                pass
            proc.wait()
    sys.stderr.write("bench.py: the run on the library's communicator %s; once more with --collective torch\n"
                     % ("did not finish in %.0f s" % limit if rc is None else "ended with code %d" % rc))
    with socket.socket() as s2:
        s2.bind(("127.0.0.1", 0))
        port2 = s2.getsockname()[1]
    cmd2 = [c if c != str(port) else str(port2) for c in cmd] + ["--collective", "torch"]
    live[0] = subprocess.Popen(cmd2, env=env, preexec_fn=own_session)
    return live[0].wait()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    import torch
    import torch.distributed as dist

    import cofactor_hip
    from cofactor_hip import dist as cdist
    from cofactor_hip import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d needs WORLD_SIZE=%d (got %d): launch it with torch.distributed.run "
                 "--nproc-per-node %d" % (args.gpus, args.gpus, world, args.gpus))
    if args.rehearse_one_gpu:
        local_rank = 0
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
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    scaling = args.scaling or ("weak" if args.rows is not None else "strong")
    if scaling == "weak":
        per_gpu = int(args.rows if args.rows is not None else args.total_rows)
        total_rows = per_gpu * world
    else:
        total_rows = int(args.total_rows)
    lo, hi = cdist.shard_bounds(total_rows, rank, world)     # rank r holds rows [lo, hi) of ONE table
    rows, n, m = hi - lo, args.num_cols, args.cat_cols
    num, cat = synth.table(torch, SEED, n, m, lo, hi, device, keys=args.keys)
    ctx = cofactor_hip.Context(local_rank)
    agg = ctx.aggregate(n, m, cofactor_hip.NB if args.nb else cofactor_hip.TRIPLE)
    num_ptrs = [t.data_ptr() for t in num]
    cat_ptrs = [t.data_ptr() for t in cat]
    torch.cuda.synchronize()
    # N > 1: the merge of the ranks' partial triples runs below the C ABI on the library's own RCCL
    # communicator; torch.distributed is the rendezvous (it carries the 128-byte id), the barrier and
    # the timing reduction.  (Two ranks on ONE device — the gloo rehearsal — cannot use RCCL.)
    comm = None
    collective = "none"
    if use_dist:
        collective = "torch.distributed all_reduce (%s)" % dist.get_backend()
        if args.collective == "lib" and not args.rehearse_one_gpu:
            # every rank tries; the ranks then agree (one tiny torch all-reduce) whether ALL of them have
            # a communicator — a rank on its own inside the library's collective would hang the job
            err = None
            # first agree that RCCL opens on every rank (the id call loads it): a rank that cannot even do
            # that must not leave the others waiting inside ncclCommInitRank
            try:
                cofactor_hip.comm_unique_id()
                probe = 1
            except Exception as e:                   # noqa: BLE001 - reported in the JSON line
                err, probe = e, 0
            okp = torch.tensor([probe], dtype=torch.int32, device=device)
            dist.all_reduce(okp, op=dist.ReduceOp.MIN)
            if int(okp.item()) == 1:
                try:
                    comm = cdist.make_comm(ctx, dist)
                except Exception as e:               # noqa: BLE001 - reported in the JSON line
                    err = e
            ok = torch.tensor([0 if comm is None else 1], dtype=torch.int32, device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                collective = "cofactor_agg_allreduce: ncclAllReduce on the library's communicator (csrc/comm.cpp)"
            else:
                if comm is not None:
                    comm.close()
                    comm = None
                collective += "; the library's communicator could not be made on every rank%s" % (
                    "" if err is None else " (this rank: %s)" % err)

    def step():
        agg.reset()
        agg.update_device_ptrs(num_ptrs, cat_ptrs, rows)
        if use_dist:
            return cdist.allreduce_triple(agg, dist, device, comm)     # ONE RCCL all-reduce, then finalize
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
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse_one_gpu else device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    value = total_rows * args.steps / dt

    # ---- value check (outside the timed region): the reduced triple against torch fp64 sums ----
    check = None
    assert int(blob[3]) == total_rows, (blob[3], total_rows)          # N of the reduced triple
    if n and not args.no_check:
        pairs = [(j, j) for j in range(n)] + [(j, (j + 1 + j % 3) % n) for j in range(n) if n > 1]
        pairs = [(min(j, k), max(j, k)) for j, k in pairs]
        lin_ref, quad_ref = reference_sums(torch, num, pairs)
        if use_dist and args.rehearse_one_gpu:
            lin_ref, quad_ref = lin_ref.cpu(), quad_ref.cpu()
        if use_dist:
            dist.all_reduce(lin_ref)
            dist.all_reduce(quad_ref)
        lin_ref, quad_ref = lin_ref.cpu().numpy(), quad_ref.cpu().numpy()
        lin = blob[4:4 + n]
        qbase = 4 + n
        if args.nb:
            pairs_chk = [(i, p) for i, p in enumerate(pairs) if p[0] == p[1]]
            qidx = lambda j, k: j
        else:
            pairs_chk = list(enumerate(pairs))
            qidx = lambda j, k: j * n - j * (j - 1) // 2 + (k - j)
        worst = max(abs(lin[k] - lin_ref[k]) / abs(lin_ref[k]) for k in range(n))
        for i, (j, k) in pairs_chk:
            worst = max(worst, abs(blob[qbase + qidx(j, k)] - quad_ref[i]) / abs(quad_ref[i]))
        assert worst < 1e-6, "reduced triple differs from the fp64 reference by %.3g relative" % worst
        check = {"N": int(blob[3]), "lin0": float(lin[0]), "quad00": float(blob[qbase]),
                 "max_rel_err_vs_torch_fp64": worst, "entries_checked": n + len(pairs_chk),
                 "lin_over_N": float(lin[0] / blob[3]), "quad00_over_N": float(blob[qbase] / blob[3])}

    if rank == 0:
        bytes_per_row = 4 * (n + m)
        cands = [("gram_kernel", prof["gram_ms"], prof["gram_launches"], 4 * n * rows),
                 ("cat_accumulate_kernel", prof["cat_ms"], prof["cat_launches"], bytes_per_row * rows),
                 (prof.get("fused_kernel", "fused_kernel"), prof["fused_ms"], prof["fused_launches"], bytes_per_row * rows)]
        kname, kms, kl, kbytes = max(cands, key=lambda c: c[1])       # the dominant kernel
        avg_ms = kms / max(1, kl)
        achieved = kbytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = "%s%s_%d_%d" % (kname, "_nb" if args.nb else "", n, m)
                # the committed PMC figure is per launch at the row count it was measured with
                if tj.get(key + "__rows") == rows:
                    traffic = tj.get(key)
            except Exception:
                traffic = None
        fname = "sum_to_%s_%d_%d" % ("nb_agg" if args.nb else "triple", n, m)
        roof = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "avg_kernel_ms": avg_ms, "launches": kl, "algorithmic_bytes_per_launch": kbytes}
        if not args.no_calibration:
            cal = calibrate(ctx)
            roof["calibrated"] = {"copy_GBs": cal["copy"], "read_GBs": cal["read"],
                                  "frac_of_copy": achieved / cal["copy"], "frac_of_read": achieved / cal["read"],
                                  "how": "float4 non-temporal copy (read + written bytes) and read-only stream, one contiguous range per workgroup, 16 workgroups per CU, "
                                         "over 4 GiB, 5 launches between HIP events (cofactor_ctx_calibrate)"}
        out = {
            "metric": "rows/sec on %s" % fname,
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s over %d rows (%d per GPU, rank r = rows [r R/G, (r+1) R/G) of one seed-%d "
                                   "table), uniform[0,1) float32 columns%s, resident in HBM" %
                                   (fname, total_rows, rows, SEED,
                                    (", %d int32 columns with %d keys" % (m, args.keys)) if m else ""),
                       "total_rows": total_rows, "rows_per_gpu": rows, "num_cols": n, "cat_cols": m,
                       "arithmetic": "f32 inputs and f32 MFMA products; f32 partial sums of at most 64 terms "
                                     "(dense) / 2048 bf16-piece terms (per-key sums) folded into f64; exact "
                                     "integer counts",
                       "parallelism": "row-sharded x%d, one RCCL all-reduce of the partial triple" % world,
                       "collective": collective},
            "roofline": roof,
            "check": check,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(num, cat, args.cpu_sample_rows, args.cpu_1t_sample_rows, n, m, args.nb)
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    agg.close()
    ctx.close()


if __name__ == "__main__":
    main()
