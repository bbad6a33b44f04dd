#!/bin/sh
# rows/s of a few categorical shapes (one JSON summary line each).  sh tests/tools/bench_shapes.sh <tag>
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
TAG=$1
run() {
  name=$1; shift
  timeout -k 10 200 python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-calibration "$@" > $R/gpurun_out/${TAG}_${name}.log 2>&1 || { echo "$name FAILED"; tail -3 $R/gpurun_out/${TAG}_${name}.log; return 0; }
  python3 -c "
import json,sys
d=json.loads(open('$R/gpurun_out/${TAG}_${name}.log').read().strip().splitlines()[-1])
r=d['roofline']
print('$name', '%.3g rows/s' % d['value'], '%.2f ms/step' % d['ms_per_step'], r['kernel'], '%.2f ms x %d' % (r['avg_kernel_ms'], r['launches']), 'frac %.3f' % r['frac'])"
}
run 10_10_k64 --total-rows 5e7 --num-cols 10 --cat-cols 10 --keys 64
run 20_20_k16 --total-rows 5e7 --num-cols 20 --cat-cols 20 --keys 16
run 10_10_k1000 --total-rows 5e7 --num-cols 10 --cat-cols 10 --keys 1000
run 4_12_k16 --total-rows 5e7 --num-cols 4 --cat-cols 12 --keys 16
run 0_10_k16 --total-rows 1e8 --num-cols 0 --cat-cols 10 --keys 16 --no-check
