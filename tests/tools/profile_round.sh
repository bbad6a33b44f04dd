#!/bin/sh
# Collects the rocprofv3 evidence kept under profiles/: kernel-trace stats and separate FETCH_SIZE /
# WRITE_SIZE passes for the 20_0 bench (default size) and the 10_10 bench.  Run on the GPU box:
#   sh tests/tools/profile_round.sh r01
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
TAG=$1
cd /tmp && export TMPDIR=/tmp
run() { # name, extra bench args...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_${name}_trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_${name}_trace.log 2>&1 || return 1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_${name}_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1 || return 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_${name}_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1 || return 1
}
run 20_0 && run 10_10 --num-cols 10 --cat-cols 10 --rows 1e8 && echo profiles collected
