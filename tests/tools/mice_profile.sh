# per-kernel times of one MICE variant: sh tests/tools/mice_profile.sh NAME [--partitioned]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
name=$1; shift
rm -rf $R/gpurun_out/$name
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$name -o t -- python3 $R/tests/tools/mice_bench.py --iterations 3 "$@" > $R/gpurun_out/$name.log 2>&1 || exit 1
grep '^{' $R/gpurun_out/$name.log | tail -1 | cut -c1-330
python3 - <<EOF
import csv,glob
f=glob.glob("$R/gpurun_out/$name/**/*kernel_stats.csv",recursive=True)[0]
open("$R/gpurun_out/$name.csv","w").write(open(f).read())
for r in csv.DictReader(open(f)):
    if "at::" in r["Name"] or "rocprim" in r["Name"]: continue
    print("%-70s calls %4s avg %9.1f us total %8.2f ms" % (r["Name"][:70],r["Calls"],float(r["AverageNs"])/1e3,float(r["TotalDurationNs"])/1e6))
EOF
rm -rf $R/gpurun_out/$name
