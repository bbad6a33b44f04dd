# per-kernel times of multiply_triple: sh tests/tools/mul_profile.sh NAME [PAIRS] [KEYS]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
name=$1; shift
rm -rf $R/gpurun_out/$name
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$name -o t -- python3 $R/tests/tools/mul_bench.py "$@" > $R/gpurun_out/$name.log 2>&1 || exit 1
grep "^multiply" $R/gpurun_out/$name.log
python3 - <<EOF
import csv,glob
f=glob.glob("$R/gpurun_out/$name/**/*kernel_stats.csv",recursive=True)[0]
open("$R/gpurun_out/$name.csv","w").write(open(f).read())
for r in csv.DictReader(open(f)):
    if "elementwise" in r["Name"] or "at::" in r["Name"]: continue
    print("%-80s calls %4s avg %10.1f us" % (r["Name"][:80],r["Calls"],float(r["AverageNs"])/1e3))
EOF
rm -rf $R/gpurun_out/$name
