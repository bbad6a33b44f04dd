#!/bin/sh
# rocprofv3 evidence for ONE bench configuration: kernel-trace stats, then separate FETCH_SIZE and
# WRITE_SIZE passes.   sh tests/tools/profile_one.sh <tag> <name> [bench args...]
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
TAG=$1; name=$2; shift 2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_${name}_trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_${name}_trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_${name}_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_${name}_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1 || exit 1
echo profile $name collected
