"""Dev tool: the one-pass kernels (COFACTOR_FUSED = 1 / 2 / 3) side by side on several shapes, plain and with a
row filter.   python tests/tools/onepass_compare.py [rows]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, os.path.join(ROOT, "duckdb-imputation_amd"))
    import torch, cofactor_hip
    rows, n, m, masked = int(float(sys.argv[2])), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    g = torch.Generator(device="cuda").manual_seed(1)
    num = [torch.rand(rows, generator=g, device="cuda") for _ in range(n)]
    cat = [torch.randint(0, 16, (rows,), generator=g, device="cuda", dtype=torch.int32) for _ in range(m)]
    msk = (torch.rand(rows, generator=g, device="cuda") < 0.9).to(torch.uint8)
    torch.cuda.synchronize()
    ctx = cofactor_hip.Context(0)
    agg = ctx.aggregate(n, m)
    upd = (lambda: agg.update_device_masked(num, cat, msk)) if masked else (lambda: agg.update_device(num, cat))
    upd(); upd(); ctx.synchronize()
    ctx.profile(True); ctx.profile_read()
    for _ in range(5):
        upd()
    ctx.synchronize()
    p = ctx.profile_read()
    k = max(("fused", "cat", "gram"), key=lambda x: p[x + "_ms"])
    print(json.dumps({"kernel": k, "ms": p[k + "_ms"] / max(1, p[k + "_launches"]) , "launches": p[k + "_launches"] // 5}))
    sys.exit(0)
rows = sys.argv[1] if len(sys.argv) > 1 else "5e7"
for (n, m) in [(10, 10), (10, 4), (4, 4), (20, 6), (2, 10), (16, 8), (6, 2)]:
    for masked in (0, 1):
        line = "%2d_%-2d %s" % (n, m, "masked" if masked else "plain ")
        for pref in ("1", "2", "3"):
            env = dict(os.environ, COFACTOR_FUSED=pref)
            out = subprocess.run([sys.executable, __file__, "--one", rows, str(n), str(m), str(masked)], env=env,
                                 capture_output=True, text=True, timeout=300)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
                line += "   pref %s: %-5s %6.3f ms x%d" % (pref, d["kernel"], d["ms"], d["launches"])
            except Exception:
                line += "   pref %s: FAILED %s" % (pref, out.stderr.strip().splitlines()[-1][:60] if out.stderr.strip() else "")
        print(line, flush=True)
