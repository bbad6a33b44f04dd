"""Per-launch averages of the SQ counters of the kernels whose name contains a substring."""
import csv, glob, json, os, sys
out, ksub = sys.argv[1], sys.argv[2]
acc = {}
for d in sys.argv[3:]:
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if ksub not in r["Kernel_Name"]:
                continue
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            acc["_vgpr"] = [float(r["VGPR_Count"])]; acc["_lds"] = [float(r["LDS_Block_Size"])]
            acc["_grid"] = [float(r["Grid_Size"])]
res = {k: sum(v) / len(v) for k, v in acc.items()}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
