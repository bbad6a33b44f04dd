# per-kernel times of the GROUP BY pool at one shape: sh tests/tools/groups_profile.sh NAME N G ROWS
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
name=$1; shift
rm -rf $R/gpurun_out/$name
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$name -o t -- python3 $R/tests/tools/groups_bench.py "$@" > $R/gpurun_out/$name.log 2>&1 || exit 1
grep "^groups" $R/gpurun_out/$name.log
python3 - <<EOF
import csv,glob
f=glob.glob("$R/gpurun_out/$name/**/*kernel_stats.csv",recursive=True)[0]
open("$R/gpurun_out/$name.csv","w").write(open(f).read())
for r in list(csv.DictReader(open(f)))[:9]: print("%-70s calls %4s avg %10.1f us" % (r["Name"][:70],r["Calls"],float(r["AverageNs"])/1e3))
EOF
rm -rf $R/gpurun_out/$name
