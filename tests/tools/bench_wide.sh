#!/bin/sh
# A/B of the per-key-sum sub-launches on wide shapes.  sh tests/tools/bench_wide.sh <tag>
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
TAG=$1
run() {
  name=$1; shift
  timeout -k 10 200 python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-calibration "$@" > $R/gpurun_out/${TAG}_${name}.log 2>&1 || { echo "$name FAILED"; tail -3 $R/gpurun_out/${TAG}_${name}.log; return 0; }
  python3 -c "
import json,sys
d=json.loads(open('$R/gpurun_out/${TAG}_${name}.log').read().strip().splitlines()[-1])
print('$name', '%.3g rows/s' % d['value'], '%.2f ms/step' % d['ms_per_step'])"
}
run 20_20_k16 --total-rows 5e7 --num-cols 20 --cat-cols 20 --keys 16
COFACTOR_NO_SUB=1 run 20_20_k16_nosub --total-rows 5e7 --num-cols 20 --cat-cols 20 --keys 16
run 20_10_k16 --total-rows 5e7 --num-cols 20 --cat-cols 10 --keys 16
COFACTOR_NO_SUB=1 run 20_10_k16_nosub --total-rows 5e7 --num-cols 20 --cat-cols 10 --keys 16
run 10_20_k16 --total-rows 5e7 --num-cols 10 --cat-cols 20 --keys 16
run 4_12_k16 --total-rows 5e7 --num-cols 4 --cat-cols 12 --keys 16
