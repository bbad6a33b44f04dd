#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/…) into the small summaries kept under profiles/.

    python profiles/summarize.py r01 gpurun_out/prof_r01 gpurun_out/pmc_fetch_r01 gpurun_out/pmc_write_r01 [tag] [traffic-key] [rows]

Writes profiles/<round>_kernel_stats[_tag].csv (kernel names cut to 100 chars),
profiles/<round>_pmc[_tag].json (per-kernel FETCH_SIZE / WRITE_SIZE per launch, raw and corrected)
and updates profiles/traffic.json (what bench.py reports as roofline.traffic).

HBM byte accounting follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced
streaming read, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is taken as is.
"""
import csv
import glob
import json
import os
import sys

HERE = os.environ.get("PROFILES_OUT") or os.path.dirname(os.path.abspath(__file__))   # PROFILES_OUT: write the summaries elsewhere


def short(name):
    name = name.replace("cofactor::(anonymous namespace)::", "")
    return name[:100]


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def main():
    rnd, prof, fetch, write = sys.argv[1:5]
    tag = ("_" + sys.argv[5]) if len(sys.argv) > 5 else ""
    key = sys.argv[6] if len(sys.argv) > 6 and sys.argv[6] else None
    nrows = int(float(sys.argv[7])) if len(sys.argv) > 7 and sys.argv[7] else None
    stats = find(prof, "_kernel_stats.csv")
    rows = list(csv.DictReader(open(stats)))
    out = os.path.join(HERE, "%s_kernel_stats%s.csv" % (rnd, tag))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        # the library's kernels first (by total time), then the harness's (torch's data generator,
        # rocBLAS dots of the value check)
        def harness(r):
            return "at::native" in r["Name"] or "rocblas" in r["Name"]
        for r in sorted(rows, key=lambda r: (1 if harness(r) else 0, -float(r["TotalDurationNs"]))):
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                        r["Percentage"], r["MinNs"], r["MaxNs"]])
    pmc = {}
    for d, counter in ((fetch, "FETCH_SIZE"), (write, "WRITE_SIZE")):
        f = find(d, "_counter_collection.csv")
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or "cofactor" not in r["Kernel_Name"]:
                continue
            k = short(r["Kernel_Name"])
            e = pmc.setdefault(k, {"vgpr": r["VGPR_Count"], "sgpr": r["SGPR_Count"],
                                   "lds_bytes": r["LDS_Block_Size"], "grid": r["Grid_Size"],
                                   "workgroup": r["Workgroup_Size"]})
            e.setdefault(counter + "_KiB_per_launch", []).append(float(r["Counter_Value"]))
    for k, e in pmc.items():
        f = e.get("FETCH_SIZE_KiB_per_launch", [])
        w = e.get("WRITE_SIZE_KiB_per_launch", [])
        fm = sum(f) / len(f) if f else 0.0
        wm = sum(w) / len(w) if w else 0.0
        e["read_bytes_per_launch_corrected"] = 2.0 * fm * 1024.0
        e["write_bytes_per_launch"] = wm * 1024.0
        e["hbm_bytes_per_launch"] = e["read_bytes_per_launch_corrected"] + e["write_bytes_per_launch"]
    with open(os.path.join(HERE, "%s_pmc%s.json" % (rnd, tag)), "w") as fh:
        json.dump(pmc, fh, indent=1, sort_keys=True)
    if key:
        tpath = os.path.join(HERE, "traffic.json")
        traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
        kname = key.rsplit("_", 2)[0]
        best = max((e for k, e in pmc.items() if kname in k), key=lambda e: e["hbm_bytes_per_launch"], default=None)
        if best:
            traffic[key] = best["hbm_bytes_per_launch"]
            traffic[key + "__source"] = "profiles/%s_pmc%s.json" % (rnd, tag)
            if nrows:
                traffic[key + "__rows"] = nrows      # bench.py reports the figure only at this size
            json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
    print(open(out).read())
    print(json.dumps(pmc, indent=1))


if __name__ == "__main__":
    main()
