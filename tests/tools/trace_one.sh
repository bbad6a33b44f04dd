#!/bin/sh
# kernel-trace stats of one bench configuration: sh tests/tools/trace_one.sh <tag> [bench args...]
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-calibration --no-check "$@" > $R/gpurun_out/${TAG}_trace.log 2>&1 || exit 1
f=$(find $R/gpurun_out/${TAG}_trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n=r["Name"].replace("cofactor::(anonymous namespace)::","").replace("cofactor::","")[:70]
    print("%-70s calls %5s avg %10.1f us  %5.1f%%" % (n, r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
